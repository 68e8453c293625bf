"""fvGP -- multi-task GP over the index set (fvgp/fvgp.py:5,480-534,626-660).

(V, Di) inputs x (V, No) outputs are flattened task-major into a single-task GP over
(V*No, Di+1) points whose last coordinate is the task index; NaN outputs are dropped.  With the
default kernel the task index is simply one more ARD dimension, so the
whole multi-task model runs on the same HIP kernels with d = Di + 1.
"""
import numpy as np

from .gp import GP


def transform_index_set(x_data, y_data, noise_variances, output_num):
    """fvGP._transform_index_set2 (fvgp/fvgp.py:626-660) without the Python double loop:
    task-major order (all points of task 0, then task 1, ...), NaN observations removed."""
    assert isinstance(x_data, np.ndarray) and isinstance(y_data, np.ndarray), "x_data / y_data must be np.ndarray"
    assert len(x_data) == len(y_data), "x_data and y_data have different lengths"
    assert y_data.shape[1] == output_num, f"y_data rows must have length output_num={output_num}"
    if noise_variances is not None:
        assert noise_variances.shape == y_data.shape, "noise_variances and y_data have different shapes"
    V = len(x_data)
    tasks = np.repeat(np.arange(output_num, dtype=np.float64), V)
    xs = np.hstack([np.tile(np.asarray(x_data, dtype=np.float64), (output_num, 1)), tasks[:, None]])
    ys = y_data.T.reshape(-1)
    keep = ~np.isnan(ys)
    nv = None if noise_variances is None else noise_variances.T.reshape(-1)[keep]
    return xs[keep], ys[keep], nv


class fvGP(GP):
    def __init__(self, x_data, y_data, init_hyperparameters=None, noise_variances=None, compute_device="gpu",
                 kernel_function=None, kernel_function_grad=None, noise_function=None, noise_function_grad=None,
                 prior_mean_function=None, prior_mean_function_grad=None, gp2Scale=False, dask_client=None,
                 gp2Scale_batch_size=10000, gp2Scale_distribution="blockwise", linalg_mode=None, ram_economy=False,
                 args=None):
        if isinstance(y_data, np.ndarray) and np.ndim(y_data) == 1:
            raise ValueError("The output number is 1, you can use the GP class for single-task GPs")
        self.output_num = y_data.shape[1]
        self.fvgp_x_data, self.fvgp_y_data, self.fvgp_noise_variances = x_data, y_data, noise_variances
        x, y, nv = transform_index_set(x_data, y_data, noise_variances, self.output_num)
        super().__init__(x, y, init_hyperparameters=init_hyperparameters, noise_variances=nv,
                         compute_device=compute_device, kernel_function=kernel_function,
                         kernel_function_grad=kernel_function_grad, noise_function=noise_function,
                         noise_function_grad=noise_function_grad, prior_mean_function=prior_mean_function,
                         prior_mean_function_grad=prior_mean_function_grad, gp2Scale=gp2Scale, dask_client=dask_client,
                         gp2Scale_batch_size=gp2Scale_batch_size, gp2Scale_distribution=gp2Scale_distribution,
                         linalg_mode=linalg_mode, ram_economy=ram_economy, args=args)
        self.x_out = np.arange(0, self.output_num)            # gp_data.set_fvgp_data, fvgp.py:534
        self.input_set_dim = self.index_set_dim - 1

    def _perform_input_checks(self, x_pred, x_out):
        super()._perform_input_checks(x_pred, x_out)
        assert x_pred.shape[1] == self.index_set_dim - 1, "wrong number of columns in x_pred"

    def update_gp_data(self, x_new, y_new, noise_variances_new=None, append=True, rank_n_update=None):
        """fvgp/fvgp.py:575-623: transform the new block to the index set, then GP.update_gp_data."""
        assert isinstance(x_new, np.ndarray), "Wrong format in x_new."
        assert isinstance(y_new, np.ndarray), "Wrong format in y_new."
        assert len(x_new) == len(y_new), "updated x and y do not have the same lengths."
        x, y, nv = transform_index_set(x_new, y_new, noise_variances_new, self.output_num)
        x_out, isd = self.x_out, self.input_set_dim
        super().update_gp_data(x, y, nv, append=append, rank_n_update=rank_n_update)
        self.x_out, self.input_set_dim = x_out, isd
        if append:                                                # the (V, Di) / (V, No) view of the data, fvgp.py:598-616
            self.fvgp_x_data = np.vstack([self.fvgp_x_data, x_new])
            self.fvgp_y_data = np.vstack([self.fvgp_y_data, y_new])
            self.fvgp_noise_variances = (None if noise_variances_new is None
                                         else np.vstack([self.fvgp_noise_variances, noise_variances_new]))
        else:
            self.fvgp_x_data, self.fvgp_y_data, self.fvgp_noise_variances = x_new, y_new, noise_variances_new
