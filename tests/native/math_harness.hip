// Host-side harness for the fixed-size algebra of csrc/ndt_math.h (the same source the device compiles):
// reads n symmetric 6x6 matrices (row-major doubles) from stdin, writes per matrix
//   lambda_min, lambda_max by sym6_extreme_eigs  |  lambda_min, lambda_max by the cyclic Jacobi it replaces;
// with the argument `s`: n angles -> (sin, cos) pairs by sincos_pose.
// Built and driven by tests/test_native_math.py (no GPU needed: only host code runs).
#include "../../ndt_feature_graph_amd/csrc/ndt_math.h"
#include <cstdio>
#include <vector>

static int sincos_mode()
{
    unsigned n = 0;
    if (fread(&n, sizeof n, 1, stdin) != 1) return 1;
    std::vector<double> in(n), out(2 * (size_t)n);
    if (fread(in.data(), sizeof(double), n, stdin) != n) return 1;
    for (unsigned k = 0; k < n; k++) sincos_pose(in[k], out[2 * k], out[2 * k + 1]);
    fwrite(out.data(), sizeof(double), out.size(), stdout);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 's') return sincos_mode();
    unsigned n = 0;
    if (fread(&n, sizeof n, 1, stdin) != 1) return 1;
    std::vector<double> in(36 * (size_t)n), out(4 * (size_t)n);
    if (fread(in.data(), sizeof(double), in.size(), stdin) != in.size()) return 1;
    for (unsigned k = 0; k < n; k++) {
        double H[6][6], A[6][6], V[6][6];
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) { H[i][j] = in[36 * k + 6 * i + j]; A[i][j] = H[i][j]; }
        double lo, hi;
        sym6_extreme_eigs(H, lo, hi);
        jacobi_static<6, false>(A, V);
        double mn = A[0][0], mx = A[0][0];
        for (int i = 1; i < 6; i++) { mn = fmin(mn, A[i][i]); mx = fmax(mx, A[i][i]); }
        out[4 * k] = lo; out[4 * k + 1] = hi; out[4 * k + 2] = mn; out[4 * k + 3] = mx;
    }
    fwrite(out.data(), sizeof(double), out.size(), stdout);
    return 0;
}
