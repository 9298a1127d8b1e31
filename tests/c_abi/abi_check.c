/* The C ABI from C (not C++, not Python): include/crabml_amd.h must compile as C99 with warnings as errors, link
 * against libcrabml_amd.so, and the host-only entry points must answer without a GPU.  Compute entry points are
 * called only to see their error path when no device is present. */
#include "crabml_amd.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

int main(void)
{
    double pi[61], Q[61 * 61], om[10], fr[4], qf[2];
    int i, j, n_ops = 0, depth = 0;
    const int32_t parents[5] = {3, 3, 4, 4, -1}, leaves[3] = {0, 1, 2};
    crab_engine *e = NULL;
    for (i = 0; i < 61; ++i) pi[i] = 1.0 / 61.0;

    if (crab_build_codon_q_matrix(2.0, 0.5, pi, 0, 0.0, Q) != CRAB_OK) return 1;
    for (i = 0; i < 61; ++i) {
        double rs = 0.0;
        for (j = 0; j < 61; ++j) rs += Q[i * 61 + j];
        if (fabs(rs) > 1e-13) return 2;
    }
    if (crab_discretise_omega(0, 0.7, 1.5, 10, om) != CRAB_OK || !(om[0] > 0.0 && om[9] < 1.0 && om[0] < om[9])) return 3;
    if (crab_model_a_quantities(2.0, 0.1, 3.0, 0.5, 0.3, pi, fr, qf) != CRAB_OK || fabs(fr[0] + fr[1] + fr[2] + fr[3] - 1.0) > 1e-15) return 4;
    if (crab_model_a_quantities(2.0, 0.1, 3.0, 0.8, 0.3, pi, fr, qf) != CRAB_ERR_ARG || !strstr(crab_last_error(NULL), "must be < 1")) return 5;
    if (crab_compile_tree(5, NULL, parents, leaves, 3, NULL, 0, &n_ops, &depth) != CRAB_OK || n_ops < 4) return 6;
    if (crab_host_pool_selftest(3, 11, 20) != CRAB_OK) return 7;

    if (!crab_device_available()) {
        /* no GPU: creation must fail loudly with CRAB_ERR_DEVICE, never fall back */
        if (crab_create(0, &e) != CRAB_ERR_DEVICE || e != NULL) return 8;
        if (!strlen(crab_last_error(NULL))) return 9;
        printf("abi ok (no device: %s)\n", crab_last_error(NULL));
    } else {
        double lnL = 0.0, bl[5] = {0.1, 0.2, 0.3, 0.1, 0.0}, one = 1.0;
        const int32_t codes[3 * 2] = {0, 5, 0, 5, 1, 64};
        if (crab_create(0, &e) != CRAB_OK) return 10;
        if (crab_set_tree(e, 5, NULL, parents, leaves, 3) != CRAB_OK) return 11;
        if (crab_set_alignment(e, codes, 3, 2, 61, 1, 0, -1) != CRAB_OK) return 12;
        /* the oracle's value for this 3-leaf case (oracle.compute_site_class_log_likelihood on the same Q, pi, tree, branch
         * lengths and codes; tests/test_c_abi.py recomputes it and compares with this constant): total lnL within 1e-10 relative */
        const double lnL_oracle = -11.394229542297818;
        if (crab_eval_site_classes(e, 1, Q, &one, pi, bl, 0, &lnL, NULL) != CRAB_OK) return 13;
        if (!(fabs(lnL - lnL_oracle) <= 1e-10 * fabs(lnL_oracle))) {
            printf("abi: device lnL %.15f differs from the oracle's %.15f\n", lnL, lnL_oracle);
            return 14;
        }
        crab_destroy(e);
        printf("abi ok (device: lnL %.12f)\n", lnL);
    }
    return 0;
}
