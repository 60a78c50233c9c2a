/* Recording stand-in for the few C-ABI entry points the glue's restart path calls (tests/test_glue_restart.py): TEST
 * INFRASTRUCTURE, linked only into tests/c/glue_restart_harness - never into the product. */
#include <stdio.h>
#include <stdlib.h>
typedef struct meso_ctx meso_ctx;
static void logf_(const char *fmt, ...);
#include <stdarg.h>
static void logf_(const char *fmt, ...)
{
    const char *p = getenv("MESO_STUB_LOG");
    if (!p) return;
    FILE *f = fopen(p, "a");
    if (!f) return;
    va_list ap;
    va_start(ap, fmt);
    vfprintf(f, fmt, ap);
    va_end(ap);
    fclose(f);
}
int meso_init(int device, meso_ctx **ctx) { *ctx = (meso_ctx *)malloc(8); logf_("meso_init %d\n", device); return 0; }
const char *meso_last_error(void) { return "stub"; }
int meso_set_mass(meso_ctx *c, int ntypes, const double *mass) { (void)c; logf_("meso_set_mass %d %.17g %.17g\n", ntypes, mass[1], mass[2]); return 0; }
int meso_pair_dpd_settings(meso_ctx *c, int style, double cut_global, int seed)
{
    (void)c; logf_("meso_pair_dpd_settings %d %.17g %d\n", style, cut_global, seed); return 0;
}
int meso_pair_dpd_coeff(meso_ctx *c, int i, int j, double a0, double gamma, double sigma, double expw, double cut)
{
    (void)c; logf_("meso_pair_dpd_coeff %d %d %.17g %.17g %.17g %.17g %.17g\n", i, j, a0, gamma, sigma, expw, cut); return 0;
}
