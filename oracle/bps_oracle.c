/* CPU ORACLE — test infrastructure, NOT product code.
 *
 * Brute-force restatement of the Basis-Point-Set encode that the reference reaches through the
 * un-vendored third-party `bps_torch` (git+https://github.com/otaheri/bps_torch, default branch,
 * no version pin; README.md:39-41) at DexGanGrasp/data/bps_encoder.py:28-55 and
 * dexgangrasp_real_robot.py:137-141:  for every basis point b_i (models/basis_point_set.npy,
 * (4096,3) f32) find the nearest cloud point  id_i = argmin_j ||b_i - x_j||^2  (strict '<', so the
 * lowest j wins a tie) and return  dist_i = sqrt(sum((x_id - b_i)^2)).
 *
 * Parity pin: the reference's own (object/ *.pcd -> bps/ *.npy) pairs, committed under
 * tests/golden/bps_real_objects.npz; tests/test_oracle_vs_golden.py checks <= 2 ulp on all 7.
 *
 * Arithmetic contract shared with the HIP kernel (dex_gan_grasp_amd/csrc/bps_encode.hip):
 *   dx = x - bx; dy = y - by; dz = z - bz;            (fp32, round-to-nearest)
 *   d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));          (fixed contraction)
 *   dist = sqrtf(d2 of the winner)                     (correctly rounded)
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off so nothing but the explicit fmaf fuses).
 */
#include <math.h>
#include <stdint.h>

void bps_encode_f32(const float *cloud, int n, const float *basis, int p, float *dists, int32_t *ids)
{
    for (int i = 0; i < p; ++i) {
        const float bx = basis[3 * i], by = basis[3 * i + 1], bz = basis[3 * i + 2];
        float best = INFINITY;
        int32_t bid = 0;
        for (int j = 0; j < n; ++j) {
            const float dx = cloud[3 * j] - bx, dy = cloud[3 * j + 1] - by, dz = cloud[3 * j + 2] - bz;
            const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            if (d2 < best) { best = d2; bid = j; }
        }
        dists[i] = sqrtf(best);
        ids[i] = bid;
    }
}

/* fp64 reference argmin + relative gap between the best and second-best squared distance
 * (where gap < 1e-6 either index is accepted). */
void bps_encode_f64(const float *cloud, int n, const float *basis, int p, double *dists, int32_t *ids,
                    int32_t *ids2, double *relgap)
{
    for (int i = 0; i < p; ++i) {
        const double bx = basis[3 * i], by = basis[3 * i + 1], bz = basis[3 * i + 2];
        double best = INFINITY, second = INFINITY;
        int32_t bid = 0, sid = -1;
        for (int j = 0; j < n; ++j) {
            const double dx = cloud[3 * j] - bx, dy = cloud[3 * j + 1] - by, dz = cloud[3 * j + 2] - bz;
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (d2 < best) { second = best; sid = bid; best = d2; bid = j; }
            else if (d2 < second) { second = d2; sid = j; }
        }
        dists[i] = sqrt(best);
        ids[i] = bid;
        ids2[i] = sid;
        relgap[i] = (n > 1 && second > 0.0) ? (second - best) / second : 1.0;
    }
}
