/* oracle_c.c -- plain C restatement of the reference's depth -> TSDF integration (CPU path).
 * TEST INFRASTRUCTURE ONLY: loaded by tests/ and by bench.py's cpu_baseline leg through ctypes
 * (oracle/oracle_c.py); never by the product.
 *
 * Restates FusionDM.fuseDepths, reference core/fusion_dm.py:180-217, one voxel at a time in the
 * reference's own operation order (compile with -ffp-contract=off: no fused multiply-add), float64
 * volumes, float64 or float32 depth.  Pinned: tests/test_oracle_c.py checks it bit for bit against
 * the numpy oracle and against the vectors produced by running the reference (tests/golden/g2, g6).
 * OpenMP over the slowest axis: voxels are independent (np.nditer order is irrelevant to the result).
 */
#include <math.h>
#include <stddef.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_c_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* returns the number of updated voxels; tsdf / tsdf_w: X*Y*Z doubles, C order (z fastest) */
long oracle_c_fuse_depths(double *tsdf, double *tsdf_w, int X, int Y, int Z, int tsdf_res, int x0, int x1,
                          const void *depth, int depth_is_f32, int H, int W, const double *K, const double *Kinv,
                          const double *lw, double scale, const double *center, double tdist, double wmax,
                          int n_threads) {
    const double c = (double)tsdf_res / 2.0;                 /* sdf_center, :183 */
    const float *d32 = (const float *)depth;
    const double *d64 = (const double *)depth;
    long count = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(static) reduction(+ : count)
#endif
    for (int x = x0; x < x1; ++x) {
        for (int y = 0; y < Y; ++y) {
            for (int z = 0; z < Z; ++z) {
                /* pos = scale*(i - c) + center                         (:188-191) */
                const double px = scale * ((double)(float)x - c) + center[0];
                const double py = scale * ((double)(float)y - c) + center[1];
                const double pz = scale * ((double)(float)z - c) + center[2];
                /* lpos = lw @ [pos,1]                                   (:193) */
                const double l0 = ((lw[0] * px + lw[1] * py) + lw[2] * pz) + lw[3];
                const double l1 = ((lw[4] * px + lw[5] * py) + lw[6] * pz) + lw[7];
                const double l2 = ((lw[8] * px + lw[9] * py) + lw[10] * pz) + lw[11];
                /* project_to_pixel(K, lpos)                             (:194, util.py:317-320) */
                const double p0 = (K[0] * l0 + K[1] * l1) + K[2] * l2;
                const double p1 = (K[3] * l0 + K[4] * l1) + K[5] * l2;
                const double p2 = (K[6] * l0 + K[7] * l1) + K[8] * l2;
                if (!(p2 != 0.0)) continue;
                const double u = p0 / p2, v = p1 / p2;
                if (!(u >= 0.0 && u < (double)(W - 1) && v >= 0.0 && v < (double)(H - 1))) continue;   /* :195 */
                const long ui = lrint(u), vi = lrint(v);         /* round half to even (:196) */
                const double dz = depth_is_f32 ? (double)d32[vi * W + ui] : d64[vi * W + ui];
                const double zd = -1.0 * dz;
                if (!(zd > 0.0)) continue;                       /* :197 */
                const double cz = (Kinv[6] * (zd * u) + Kinv[7] * (zd * v)) + Kinv[8] * (zd * 1.0);  /* :198-200 */
                const double sd = cz - l2;                       /* :201 */
                if (!(sd > -1.0 * tdist)) continue;              /* :203 */
                const size_t i = ((size_t)x * Y + y) * Z + z;
                const double wt = tsdf_w[i];
                const double m = sd < tdist ? sd : tdist;
                tsdf[i] = (scale * tsdf[i] * wt + m * 1.0) / (scale * (1.0 + wt));     /* :209 */
                const double nw = 1.0 + wt;
                tsdf_w[i] = nw < wmax ? nw : wmax;               /* :210 */
                ++count;
            }
        }
    }
    return count;
}
