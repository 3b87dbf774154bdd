/* A plain C99 client of include/wtphys.h -- what a cgo / JNI / FFI binding of the boundary looks like from the
 * other side: plain pointers and sizes, no C++, no torch.  Reads one binary request
 *   int64 N, n, steps; double dt; par[WT_NP][N]; bc[WT_NB][N]; pH[N][n]; Cl[N][n]; T[N][n]
 * advances the ensemble, writes pH, Cl, T [N][n], time [N], status [N] (as doubles) to the reply file.
 * tests/test_host_api.py compiles it (gcc -std=c99 -pedantic -Werror: the header is valid C, every symbol it uses
 * links); tests/test_gpu_parity.py runs it and compares with the ctypes path bit for bit. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "wtphys.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != WT_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, wt_last_error()); return 2; } \
    } while (0)

static double *rd(FILE *f, size_t count)
{
    double *p = (double *)malloc(count * sizeof(double));
    if (!p || fread(p, sizeof(double), count, f) != count) { fprintf(stderr, "short request\n"); exit(3); }
    return p;
}

int main(int argc, char **argv)
{
    int64_t hdr[3];
    double dt;
    FILE *in, *out;
    wt_ensemble *h = NULL;
    if (argc != 3) { fprintf(stderr, "usage: abi_client request.bin reply.bin\n"); return 1; }
    if (wt_abi_version() != WT_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    in = fopen(argv[1], "rb");
    if (!in || fread(hdr, sizeof(int64_t), 3, in) != 3 || fread(&dt, sizeof(double), 1, in) != 1) return 1;
    {
        const size_t N = (size_t)hdr[0], n = (size_t)hdr[1];
        const int steps = (int)hdr[2];
        double *par = rd(in, WT_NP * N), *bc = rd(in, WT_NB * N);
        double *pH = rd(in, N * n), *Cl = rd(in, N * n), *T = rd(in, N * n);
        double *time = (double *)malloc(N * sizeof(double)), *st_d = (double *)malloc(N * sizeof(double));
        uint32_t *status = (uint32_t *)malloc(N * sizeof(uint32_t));
        size_t i;
        fclose(in);
        CHECK(wt_ensemble_create((int64_t)N, (int)n, 0, par, &h));
        CHECK(wt_ensemble_set_state(h, pH, Cl, T, NULL));
        CHECK(wt_ensemble_set_boundary(h, bc));
        CHECK(wt_ensemble_step(h, dt, steps, 1));
        CHECK(wt_ensemble_synchronize(h));
        CHECK(wt_ensemble_get_state(h, pH, Cl, T, time, NULL));
        CHECK(wt_ensemble_get_status(h, status));
        CHECK(wt_ensemble_destroy(h));
        for (i = 0; i < N; ++i) st_d[i] = (double)status[i];
        out = fopen(argv[2], "wb");
        if (!out) return 1;
        fwrite(pH, sizeof(double), N * n, out); fwrite(Cl, sizeof(double), N * n, out); fwrite(T, sizeof(double), N * n, out);
        fwrite(time, sizeof(double), N, out); fwrite(st_d, sizeof(double), N, out);
        fclose(out);
        free(par); free(bc); free(pH); free(Cl); free(T); free(time); free(st_d); free(status);
    }
    return 0;
}
