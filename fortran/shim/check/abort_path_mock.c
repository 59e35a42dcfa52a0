/* Stands in for a few entry points of libseaice_hip in `make check-abort-path` (defined in the executable, they take
 * precedence over the library's): the remap reports the reference's run-time abort without any device. */
#include "seaice_hip.h"

static const char *msg = "";
int sih_ir_run(sih_ctx *ctx, double dt, int checks)
{
    (void)ctx; (void)dt; (void)checks;
    msg = "IR: Negative mass in IR";
    return SIH_ERR_IR;
}
const char *sih_last_error(void) { return msg; }

/* the refresh of the pool fields around the remap: nothing to move here */
size_t sih_field_count(sih_ctx *ctx, const char *name, int *elemSize)
{
    (void)ctx; (void)name;
    if (elemSize) *elemSize = 8;
    return 3;
}
int sih_fields_put(sih_ctx *ctx, int n, const char *const *names, const void *const *hosts, const size_t *counts)
{
    (void)ctx; (void)n; (void)names; (void)hosts; (void)counts;
    return SIH_OK;
}
int sih_fields_get(sih_ctx *ctx, int n, const char *const *names, void *const *hosts, const size_t *counts)
{
    (void)ctx; (void)n; (void)names; (void)hosts; (void)counts;
    return SIH_OK;
}
