/* A host written in plain C against include/meso_hip.h - what a cgo / JNI / LAMMPS binding does, without Python or ctypes
 * in between.  Reads a deck (n, box, x, v as raw little-endian doubles), runs the dp.run settings for nsteps through the C
 * ABI (the glue's call sequence: INTEGRATION.md section 4) and writes tag-ordered x, v, f plus the temperature.
 * Built and run by tests/test_gpu_cabi_host.py:  gcc abi_smoke.c -I include -L meso_amd -lmeso_hip  */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "meso_hip.h"

#define CK(call)                                                                   \
    do {                                                                           \
        int rc_ = (call);                                                          \
        if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, meso_last_error()); return 10 + rc_; } \
    } while (0)

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: abi_smoke IN OUT NSTEPS STYLE(0|1)\n"); return 2; }
    FILE *in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 3; }
    long long n64;
    double lo[3], hi[3];
    if (fread(&n64, 8, 1, in) != 1 || fread(lo, 8, 3, in) != 3 || fread(hi, 8, 3, in) != 3) return 3;
    int n = (int)n64, nsteps = atoi(argv[3]), style = atoi(argv[4]);
    double *x = malloc(sizeof(double) * 3 * n), *v = malloc(sizeof(double) * 3 * n), *f = malloc(sizeof(double) * 3 * n);
    int *tag = malloc(sizeof(int) * n), *type = malloc(sizeof(int) * n);
    if (fread(x, 8, 3 * (size_t)n, in) != 3 * (size_t)n || fread(v, 8, 3 * (size_t)n, in) != 3 * (size_t)n) return 3;
    fclose(in);
    for (int i = 0; i < n; i++) { tag[i] = i + 1; type[i] = 1; }

    meso_ctx *c = NULL;
    const int periodic[3] = {1, 1, 1};
    const double mass[2] = {0.0, 1.0};
    CK(meso_init(0, &c));
    CK(meso_set_box(c, lo, hi, periodic));
    CK(meso_set_mass(c, 1, mass));
    CK(meso_atoms_upload(c, n, x, v, tag, type, NULL, NULL));
    CK(meso_neighbor(c, 0.3, 5, 0, 0));                                   /* neighbor 0.3 bin; neigh_modify delay 0 every 5 check no */
    CK(meso_timestep(c, 0.005));
    CK(meso_pair_dpd_settings(c, style ? MESO_PAIR_DPD_FAST : MESO_PAIR_DPD, 1.0, 419084618));
    CK(meso_pair_dpd_coeff(c, 1, 1, 15.0, 4.5, 3.0, 1.0, 1.0));
    CK(meso_setup(c));
    CK(meso_run(c, nsteps));
    double T = 0.0;
    CK(meso_compute_temp(c, &T));
    int nl = 0, ng = 0, nb = 0;
    CK(meso_atoms_count(c, &nl, &ng, &nb));
    if (nl != n) { fprintf(stderr, "atom count changed: %d != %d\n", nl, n); return 4; }
    int *otag = malloc(sizeof(int) * n);
    double *ox = malloc(sizeof(double) * 3 * n), *ov = malloc(sizeof(double) * 3 * n), *of = malloc(sizeof(double) * 3 * n);
    CK(meso_atoms_download(c, ox, ov, of, otag, NULL, NULL));             /* device order */
    for (int i = 0; i < n; i++) {
        int t = otag[i] - 1;
        memcpy(x + 3 * t, ox + 3 * i, 24); memcpy(v + 3 * t, ov + 3 * i, 24); memcpy(f + 3 * t, of + 3 * i, 24);
    }
    /* an error must come back as a status with a message, never as an abort (util_meso.h:96-97 raises SIGABRT) */
    if (meso_pair_dpd_coeff(c, 7, 7, 1.0, 1.0, 1.0, 1.0, 1.0) == MESO_OK || !meso_last_error()[0]) return 5;
    CK(meso_finalize(c));
    FILE *out = fopen(argv[2], "wb");
    if (!out) return 3;
    fwrite(&T, 8, 1, out);
    fwrite(x, 8, 3 * (size_t)n, out); fwrite(v, 8, 3 * (size_t)n, out); fwrite(f, 8, 3 * (size_t)n, out);
    fclose(out);
    printf("abi_smoke: n=%d steps=%d ghosts=%d T=%.6f\n", n, nsteps, ng, T);
    return 0;
}
