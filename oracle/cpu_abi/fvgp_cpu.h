/* shared by the two translation units of the CPU twin of the C ABI (TEST INFRASTRUCTURE ONLY, see fvgp_cpu.c) */
#ifndef FVGP_CPU_H
#define FVGP_CPU_H
#include <stdint.h>
#include "../../include/fvgp_hip.h"
struct fvgp_handle {
    int device; int64_t outer_block;
    fvgp_collectives coll; int coll_rank, coll_nranks;       /* row-sharded evaluation: the caller's collectives (gloo in the tests) */
};
#ifdef __cplusplus
extern "C" {
#endif
void fvgp_cpu_set_err(const char *s);
#ifdef __cplusplus
}
#endif
#endif
