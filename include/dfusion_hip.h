/* dfusion_hip.h -- C ABI of libdfusion_hip.so, the MI355X (gfx950) implementation of the
 * per-frame DynamicFusion hot path of nintendops/DynamicFusion_Body.
 *
 * The reference has no FFI: its own device plug-in is a Python subclass overriding one
 * numpy-in/numpy-out method (class FusionDM_GPU, core/fusion_dm.py:563-574,600).  This
 * library is what such an override binds through ctypes; every entry
 * point names the reference function whose arithmetic it reproduces.
 *
 * Conventions
 *  - All array pointers are DEVICE pointers (hipMalloc / torch.Tensor.data_ptr()) unless
 *    the parameter is a small fixed-size `const double[...]`, which is HOST memory read
 *    during the call (mask arithmetic is fp64, so small matrices travel as doubles).
 *  - Volumes are C-ordered [x][y][z], z fastest (np.nditer order, core/fusion_dm.py:186;
 *    the OpenCL kernel's idx = x*RES_Z*RES_Y + y*RES_Z + z, :637).  A volume buffer holds
 *    the axis-0 planes [x0, x1) of a res[0] x res[1] x res[2] grid (slab partition across
 *    GPUs); voxel indices used in the arithmetic are always GLOBAL.
 *  - `vol_dtype` / `depth_dtype`: DFH_F32 or DFH_F64.  fp32 volumes are the product
 *    layout (16 B/voxel read-modify-write); fp64 volumes reproduce the reference's float64
 *    arrays bit for bit and exist for parity checking.
 *  - Calls are asynchronous on `stream` (a hipStream_t; NULL = default stream).
 *  - Return value: 0 on success, <0 on error (DFH_E_*); dfh_last_error() describes the
 *    last failure on the calling thread.  Nothing throws across the ABI.
 */
#ifndef DFUSION_HIP_H
#define DFUSION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFH_ABI_VERSION 4

#define DFH_F32 0
#define DFH_F64 1

#define DFH_OK 0
#define DFH_E_BADARG (-1)
#define DFH_E_HIP (-2)
#define DFH_E_UNSUPPORTED (-3)
#define DFH_E_TIMEOUT (-4)

/* ABI version of the loaded library (== DFH_ABI_VERSION of the header it was built from). */
int dfh_version(void);

/* Message for the last non-zero return on this thread ("" if none). */
const char *dfh_last_error(void);

/* Blocks until `stream` has drained (hipStreamSynchronize). */
int dfh_stream_synchronize(void *stream);

/* Development switches of the library (A/B experiments, tests that force a fall-back path).  They are NOT read from the
 * environment on the call paths: the table is filled once, at the first call into the library, from the single environment
 * variable DFH_OPTIONS="name=value,name=value" and is changed afterwards only through dfh_set_option().  Names are listed in
 * tools/README.md (e.g. "k1_no_bricks", "pcg_multilaunch", "pcg_spin_limit"); a value of -1 means "unset / library default".
 * dfh_set_option returns DFH_E_BADARG for an unknown name; dfh_get_option returns the current value (LONG_MIN if unknown).
 * No reference counterpart (the reference has no tuning switches). */
int dfh_set_option(const char *name, long value);
long dfh_get_option(const char *name);

/* A1  FusionDM.fuseDepths(dm, lw, tsdf, tsdf_w, scale, center, wmax)  core/fusion_dm.py:180-217
 * (CPU-path semantics; the OpenCL variant :600-737 is NOT what is reproduced).
 * For every voxel i=(x,y,z), x in [x0,x1):
 *   pos  = scale*(i - tsdf_res/2) + center                       (:183,:191)
 *   lpos = lw*[pos,1];  (u,v) = (K*lpos)_{0,1}/(K*lpos)_2, skipped if (K*lpos)_2 == 0   (:193-194)
 *   visible iff 0<=u<W-1 and 0<=v<H-1                            (:195)
 *   z = -depth[rint(v)][rint(u)] (round-half-even), valid iff z>0   (:196-197)
 *   sd = (Kinv*(z*[u,v,1]))_2 - lpos_2;  update iff sd > -tdist   (:198-203)
 *   T <- (scale*T*w + min(tdist,sd)) / (scale*(1+w));  w <- min(1+w, wmax)   (:209-210)
 * tsdf/tsdf_w: planes [x0,x1) of the volume, dtype vol_dtype.  depth: H x W row-major,
 * negative depths, 0 = no measurement, dtype depth_dtype.  K, Kinv: 3x3 row-major;
 * lw: 3x4 row-major; center: 3.  tsdf_res is the ctor's tsdf_res (:60,:183).
 * workspace (may be NULL): device scratch of dfh_integrate_workspace_bytes(1, H, W, res, x0, x1) bytes.  With it, float32
 * volumes are swept in 4 x 4 x 16 voxel bricks after a classification pass (same call, same stream) that marks the bricks
 * whose eight projected corners prove that the view updates none of their voxels -- outside the image, or behind the
 * surface by more than tdist according to a max-depth pyramid of the depth map -- and the sweep skips those: same result,
 * bit for bit, about half the projection work for a typical view.  Without it every voxel is projected. */
size_t dfh_integrate_workspace_bytes(int n_views, int H, int W, const int res[3], int x0, int x1);
int dfh_integrate_depth(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3], int tsdf_res,
                        int x0, int x1, const void *depth, int depth_dtype, int H, int W,
                        const double K[9], const double Kinv[9], const double lw[12],
                        double scale, const double center[3], double tdist, double wmax,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Which sweep dfh_integrate_depth takes for float32 volumes of this slab and depth-map size (no launch): one of
 * DFH_K1_PATH_*.  All of them produce the same volumes bit for bit; they differ in the bytes they move (measurement code
 * counts those of the path taken).  have_workspace: a workspace of dfh_integrate_workspace_bytes(1, ...) bytes will be passed.
 * No reference counterpart. */
#define DFH_K1_PATH_EXACT 0          /* every voxel through the reference's fp64 chain (fp64 volumes, oversized depth maps) */
#define DFH_K1_PATH_ROWS 1           /* one 1-KiB z run per wave; T, w loaded and stored for updated 16-byte packs only */
#define DFH_K1_PATH_COLUMNS 2        /* 4 x 2 x 32 bricks, a wave walks a column of them; T, w of every pack loaded, updated packs stored */
#define DFH_K1_PATH_COLUMNS_CULLED 3 /* the same behind a depth pyramid + brick classification: bricks no voxel of which can be updated are skipped */
int dfh_integrate_depth_path(int vol_dtype, const int res[3], int x0, int x1, int H, int W, int have_workspace);

/* The same for n_views depth maps in ONE sweep of the volume: what the reference's loops over fuseDepths do
 * (core/fusion_dm.py:152-154 initial fusion, :166-170 compute_live_tsdf), with every voxel's T and w read once, updated
 * view by view in registers -- the float32 operations of consecutive dfh_integrate_depth calls, so the same bits --
 * and written once.  depth: HOST array of n_views device pointers (all H x W, depth_dtype); lw: n_views x 12.
 * n_views <= 16.  workspace: device scratch; dfh_integrate_workspace_bytes(n_views, H, W, res, x0, x1) bytes enable the brick
 * sweep with the per-(brick, view) classification described above (a brick runs only the views that may update it, a brick
 * no view updates is never loaded), dfh_integrate_multi_workspace_bytes(n_views)
 * bytes (the views' folded projection parameters only) the plain sweep; without a workspace, for float64 volumes and for
 * depth maps beyond 2048 pixels a side the call runs one sweep per view. */
size_t dfh_integrate_multi_workspace_bytes(int n_views);
int dfh_integrate_depth_multi(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3], int tsdf_res, int x0, int x1,
                              int n_views, const void *const *depth, int depth_dtype, int H, int W,
                              const double K[9], const double Kinv[9], const double *lw, double scale,
                              const double center[3], double tdist, double wmax, void *workspace,
                              size_t workspace_bytes, void *stream);

/* A live volume from scratch: the volumes are first set to (fresh_value, 0) -- the reference's np.zeros(...) + tdist and
 * np.zeros(...) in front of its fuseDepths loops, core/fusion_dm.py:100-101,152-153 -- and the n_views depth maps are then fused as
 * by dfh_integrate_depth_multi: the result is that of the two fills followed by that call, bit for bit (fresh_value is rounded to
 * the volume's type).  With the column sweep the fill is part of the sweep: nothing is read and every voxel of the slab is written
 * once (a 256^3 live volume of three views: fills 38 + sweep 89 -> sweep 84 us); otherwise the slab is filled by a launch of its own first.
 * n_views == 0 only fills. */
int dfh_integrate_depth_multi_fresh(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3], int tsdf_res, int x0, int x1,
                                    double fresh_value, int n_views, const void *const *depth, int depth_dtype, int H, int W,
                                    const double K[9], const double Kinv[9], const double *lw, double scale,
                                    const double center[3], double tdist, double wmax, void *workspace,
                                    size_t workspace_bytes, void *stream);

/* A2 (optional)  the arithmetic of the reference's OpenCL kernel `fuse_depth`  core/fusion_dm.py:630-674 -- NOT that of the CPU
 * path above: index -> pixel through one float32 3x4 map proj = K lw IND (:640-646,:695), bilinear depth (:605-622), pixels without
 * or with near depth (pz <= tdist) carve free space (dz = -tdist, :652-653), dz = voxel depth - measured depth (:655-658), update
 * iff dz < tdist: w' = min(1 + w, wmax), T <- ((w' - 1) T + max(-tdist, dz)) / w', w <- w' (:667-672).  All float32, in the kernel
 * text's operation order, no contraction.  tsdf / tsdf_w: float32 planes [x0,x1), updated in place (the reference's host code
 * copies its inputs first, :690-691); depth float32 H x W; tdist / wmax as the float literals the reference bakes in ("%ff" of
 * the Python values, :682-687).  A pixel coordinate that is NaN (w == 0) is skipped (undefined in the reference). */
int dfh_integrate_depth_ocl(float *tsdf, float *tsdf_w, const int res[3], int x0, int x1, const float *depth, int H, int W,
                            const float proj[12], const float kinv_row2[3], float tdist, float wmax, void *stream);

/* A3  FusionDM.updateTSDF(curr_tsdf, wmax)                     core/fusion_dm.py:300-316
 * For every canonical voxel i, x in [x0,x1):
 *   q = dqb_warp(lw_dq, float32(i))          (core/util.py:68-72; lw_dq = `_lw`, 8 doubles, voxel-index
 *                                             space, may be non-unit after solve, fusion_dm.py:282)
 *   s = interpolate_tsdf(q, live)            (core/util.py:102-137: None outside [0,R-1]^3, ceil() upper
 *                                             corner, y/z fractions swapped -- reproduced)
 *   update iff s is not None and s > -tdist: T <- (T*w + min(tdist,s))/(1+w); w <- min(1+w, wmax)
 * live: the full live volume live_res[0] x live_res[1] x live_res[2] (every rank holds all of it: the warp
 * gathers across slab boundaries), dtype live_dtype. */
int dfh_fuse_volume_rigid(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3], int x0, int x1,
                          const void *live, int live_dtype, const int live_res[3],
                          const double lw_dq[8], double tdist, double wmax, void *stream);

/* A4-A6  Fusion.updateTSDF(curr_tsdf, wmax)                    core/fusion.py:153-198
 * For every canonical voxel i, x in [x0,x1):
 *   loc = the knn nearest nodes of float32(i), nearest first       (KDTree.query(pos,k=knn+1)[1][:-1], :175-176)
 *   b   = sum_j exp(-(|i - v_j| / (2*w_j))^2) * dq_j ; b^ = b/|b|_8 (identity if |b|_8 == 0)   (dq_blend, :527-551)
 *   q   = dqb_warp(lw_dq, float32(dqb_warp(b^, i)))                (warp, :502-520; util.py:69)
 *   s   = interpolate_tsdf(q, live); update iff s is not None and s > -tdist              (:178-179)
 *   wi  = sum_j |v_j - i| / knn ; wt = w, or wi if w == 0                                 (:180-187)
 *   T <- (T*wt + min(tdist,s)*wi)/(wi + wt) ; w <- min(wi + wt, wmax)                      (:189-190)
 * node_pos: n_nodes x 3, node_dq: n_nodes x 8, node_w: n_nodes (the nodes' 4th tuple entry, 2*radius,
 * :116), all device fp64.  1 <= knn <= 8.  workspace: device scratch of dfh_dqb_workspace_bytes() bytes
 * holding per-brick candidate node lists; they depend only on (node_pos, knn, grid, slab) and are rebuilt
 * when rebuild_candidates != 0.  A workspace of dfh_dqb_workspace_bytes_cached() bytes (16-byte aligned; the
 * plain size when n_nodes > 65536) additionally keeps per voxel the knn node indices (level 1: 2*knn bytes) and
 * the blend weights exp(..) and wi (level 2: + 8*(knn+1) bytes): the call with rebuild_candidates != 0 stores
 * them, later calls skip the node search and the sqrt/divide/exp chain -- same results, bit for bit, since
 * these values depend on (node_pos, node_w, knn, grid, slab) and not on node_dq.  The level in use is
 * inferred from workspace_bytes.
 * float32 volumes, knn = 4, level-2 workspace, lw_dq the identity, steady state (round 4): voxels that provably sample only
 * live voxels holding exactly tdist -- a per-brick bound on |warp(i) - i| from the brick's candidate nodes' DQs, a per-cell
 * "all 64 live voxels == tdist" mask -- skip the warp: s = tdist whatever the position; same bits (dfh_dqb_skip_layout;
 * option k3_skip = 0 switches it off). */
size_t dfh_dqb_workspace_bytes(const int res[3], int x0, int x1);
size_t dfh_dqb_workspace_bytes_cached(const int res[3], int x0, int x1, int knn, int n_nodes, int level);
int dfh_fuse_volume_dqb(void *tsdf, void *tsdf_w, int vol_dtype, const int res[3], int x0, int x1,
                        const void *live, int live_dtype, const int live_res[3],
                        const double *node_pos, const double *node_dq, const double *node_w, int n_nodes,
                        int knn, const double lw_dq[8], double tdist, double wmax,
                        void *workspace, size_t workspace_bytes, int rebuild_candidates, void *stream);

/* ---- warp-field solve ------------------------------------------------------------------------------
 * All arrays device fp64 unless noted; point / normal / node arrays are row-major (n x 3, n x 8).
 *
 * A9   FusionDM.computef_lw(x)                                   core/fusion_dm.py:285-297
 *   out[i] = dqb_warp_normal(x, normals[i]) . (dqb_warp(x, verts[i]) - corr[i])
 *   (verts/normals already restricted to `_corridx`; row i pairs with corr[i]). */
int dfh_residual_rigid(const double *verts, const double *normals, const double *corr, int n, const double x[8],
                       double *out, void *stream);

/* Rigid Gauss-Newton normal equations of 0.5*|computef_lw(x)|^2 for the left twist on x:
 * out44[0..35] = J^T J (6x6 row-major), out44[36..41] = J^T r, out44[42] = 0.5|r|^2, out44[43] = rows
 * used.  valid (uint8 per row) may be NULL. */
int dfh_gn_build_rigid(const double *verts, const double *normals, const double *corr, const unsigned char *valid, int n,
                       const double x[8], double *out44, void *stream);

/* A10  data rows of Fusion.computef(x, tdw, trw, rw) / Fusion.computef_lw   core/fusion.py:444-473
 *   (x', n') = warp(verts[s], node_dq[nbr[s]], nbr[s], normals[s], m_lw=lw_dq)   (:502-520)
 *   out[s] = n' . (x' - corr[s]);  nbr: n_verts x knn int32 (`_neighbor_look_up`, :121-123). */
int dfh_residual_data(const double *verts, const double *normals, const double *corr, const int *nbr, int n_verts,
                      int knn, const double *node_dq, const double *node_pos, const double *node_w, int n_nodes,
                      const double lw_dq[8], double *out, void *stream);

/* A10  regularisation rows of Fusion.computef                      core/fusion.py:475-484
 *   node_nbr[i][j] = _neighbor_look_up[_nodes[i][0]][j]  (n_nodes x knn int32)
 *   out[(i*knn + j)*3 + c] = rw * max(w_i, w_j) * (dqb_warp(dq_i, v_j) - dqb_warp(dq_j, v_j))[c]. */
int dfh_residual_reg(const int *node_nbr, int n_nodes, int knn, const double *node_dq, const double *node_pos,
                     const double *node_w, double rw, double *out, void *stream);

/* A5 for a batch: Fusion.warp(v, dqs[nbr], nbr, normal=n, m_lw=lw_dq) (core/fusion.py:502-520) for every
 * vertex; nbr == NULL applies only lw_dq (dqb_warp(_lw, v), dqb_warp_normal(_lw, n), fusion_dm.py:230-231).
 * normals / out_nrm may both be NULL. */
int dfh_warp_points(const double *verts, const double *normals, const int *nbr, int n_verts, int knn, const double *node_dq,
                    const double *node_pos, const double *node_w, int n_nodes, const double lw_dq[8], double *out_pos,
                    double *out_nrm, void *stream);

/* The selection loop of setupCorrespondences (core/fusion_dm.py:229-244; core/fusion.py:258-276, 'clpts'):
 * for every warped vertex the knn nearest live vertices (nearest first), best = the first with the smallest
 * cost |wn . (vp - p)| below the initial best_cost 1 (else the nearest), keep = best_cost <= tolerance.
 * corr_out n x 3, cost_out n (may be NULL), keep_out n uint8. */
int dfh_closest_correspondences(const double *warped_pos, const double *warped_nrm, int n_verts, const double *live_verts,
                                int n_live, int knn, double tolerance, double *corr_out, double *cost_out,
                                unsigned char *keep_out, void *stream);

/* k nearest nodes (nearest first; KDTree.query order, core/fusion.py:121-123) and the Gaussian DQB
 * weights exp(-(|p - v_j| / (2 w_j))^2) (:537) of arbitrary sample points.  Both are static while the
 * graph is unchanged.  nbr_out: n_samples x knn int32; weights_out: n_samples x knn. */
int dfh_sample_knn(const double *sample_pos, int n_samples, const double *node_pos, const double *node_w, int n_nodes,
                   int knn, int *nbr_out, double *weights_out, void *stream);
/* Where the constant-live skip of dfh_fuse_volume_dqb (float32 volumes, knn = 4, stored neighbourhoods, m_lw = identity; round 4)
 * keeps its per-call tables inside a level-2 workspace, as byte offsets from the workspace's start: out[0] live-cell mask U,
 * [1] slab-cell mask S, [2] per-brick reach (uint8: 1 / 2 cells, 255 = no bound), [3] per-brick displacement bound (float32,
 * voxels; -1 = not computed), [1] is one byte per brick (1 = constant-live stream, 0 = warp kernel), [4], [5] unused; [6..8] live cells
 * along x, y and 64-bit words per cell row, [9..10] slab cell rows; [11] 1 when these sizes admit the skip at all; [12] per brick
 * the 16 node ids (uint16, ascending, 0xffff = none, first = 0xfffe: too many) its voxels blend.  For tests and measurement code:
 * the proof obligation "no voxel moves further than its brick's bound" is checked against [3] (tests/test_gpu_fuse_volume.py).
 * No reference counterpart. */
int dfh_dqb_skip_layout(const int res[3], int x0, int x1, const int live_res[3], int knn, int n_nodes, size_t out[13]);
/* The same through the per-brick candidate lists of a dfh_fuse_volume_dqb workspace (built for the same node_pos, knn,
 * grid and slab by dfh_dqb_build_candidates or by a dfh_fuse_volume_dqb call with rebuild_candidates != 0): a point
 * scans the list of the brick of its nearest voxel centre (the lists carry the head-room that makes this exact for
 * off-lattice points); points outside the slab's lattice and bricks whose list overflowed scan every node.
 * Same output as dfh_sample_knn, bit for bit. */
int dfh_dqb_build_candidates(const int res[3], int x0, int x1, const double *node_pos, int n_nodes, int knn,
                             void *workspace, size_t workspace_bytes, void *stream);
int dfh_sample_knn_bricks(const double *sample_pos, int n_samples, const double *node_pos, const double *node_w,
                          int n_nodes, int knn, const int res[3], int x0, int x1, const void *workspace,
                          size_t workspace_bytes, int *nbr_out, double *weights_out, void *stream);

/* ---- deformation-graph maintenance: the device side of Fusion.update_graph / construct_graph ---------------------
 * (core/fusion.py:101-123, 201-239; the greedy radius subsampling of the unsupported vertices, core/util.py:27-47, is
 * sequential by definition and stays with the caller).
 * dfh_nearest_points: idx_out[q] = nearest cloud point of query q (KDTree(cloud).query(q), :209-212 -- a node's anchor
 *   vertex; ties go to the lower index), d2_out (may be NULL) its squared distance.
 * dfh_graph_unsupported: flag_out[v] = 1 iff min over the vertex's knn nodes nbr[v][.] of |node - v| / node_w >= 1
 *   (the "unsupported surface point" test, :215-219).
 * dfh_dq_blend_points: dq_out[p] = Fusion.dq_blend(points[p]) over the nodes nbr[p][.] (:527-551; the DQ a newly
 *   inserted node starts from, :222), 8 doubles per point. */
int dfh_nearest_points(const double *query, int n_query, const double *cloud, int n_cloud, int *idx_out, double *d2_out, void *stream);
int dfh_graph_unsupported(const double *verts, int n_verts, const int *nbr, int knn, const double *node_pos, const double *node_w,
                          int n_nodes, unsigned char *flag_out, void *stream);
int dfh_dq_blend_points(const double *points, int n_points, const int *nbr, int knn, const double *node_dq, const double *node_pos,
                        const double *node_w, int n_nodes, double *dq_out, void *stream);

/* Projective data association (not in the reference, which matches marching-cubes vertices through a
 * KD-tree, core/fusion.py:255-276): warp each sample with the current field (Fusion.warp), map index ->
 * world (pos = scale*(i - half) + center, fusion_dm.py:191) -> camera (lw_cam, :193) -> pixel (:194-195),
 * take the nearest depth pixel z = -depth[rint(v)][rint(u)] (:196), back-project K^-1 (z [u,v,1])
 * (:198-200) and map back to index space.  valid_out[s] = 0 when outside the image, no depth, or farther
 * than max_dist voxels from the warped sample (max_dist <= 0: no gate). */
/* out[i] = in[order[i]] for the four per-sample arrays at once (samples are sorted by node tuple before the build:
 * few runs per tile of dfh_gn_tile_samples() samples).  order: n_samples int64 indices, a permutation. */
int dfh_permute_samples(const long *order, int n_samples, int knn, const double *pos, const double *nrm, const int *nbr,
                        const double *weights, double *pos_out, double *nrm_out, int *nbr_out, double *weights_out, void *stream);

int dfh_gn_associate(const double *sample_pos, const int *nbr, const double *weights, int n_samples, int knn,
                     const double *node_dq, const double lw_dq[8], const void *depth, int depth_dtype, int H, int W,
                     const double K[9], const double Kinv[9], const double lw_cam[12], double scale,
                     const double center[3], double half, double max_dist, double *corr_out,
                     unsigned char *valid_out, void *stream);

/* Association against SEVERAL live views (BASELINE config 5: the live frame is eight depth maps).  Every view is tried in
 * turn with the arithmetic of dfh_gn_associate; a sample keeps the correspondence of the view in which it lies closest to
 * the observed surface (smallest |c - x'| among the views where it is valid; max_dist gates every view; ties go to the lower
 * view index), so it still contributes ONE data row and the block pattern / plan do not depend on the number of views.  One
 * view gives dfh_gn_associate's result bit for bit.  No reference counterpart (its correspondences are mesh-to-mesh,
 * core/fusion.py:255-276; its view loop is the TSDF fusion's, core/fusion_dm.py:166-170); oracle: gn_np.associate_depth_views.
 *   dfh_gn_pack_views writes the views' table (extrinsics, their inverses, depth pointers) into `views_out`, a device buffer
 *   of dfh_gn_views_bytes(n_views) bytes: once per frame; depth[v]: device pointers to H x W maps of one dtype; lw_cam: 12
 *   doubles per view (host).  The *_views calls take that table instead of (depth, lw_cam). */
#define DFH_GN_MAX_VIEWS 16
size_t dfh_gn_views_bytes(int n_views);
int dfh_gn_pack_views(void *views_out, int n_views, const void *const *depth, const double *lw_cam, void *stream);
/* The same table with, behind it, per view a table of 16 x 16-pixel cells {smallest, largest valid z = -depth} of FLOAT32 depth
 * maps (views_out: dfh_gn_views_bytes_cells(n_views, H, W) bytes).  With it the fused builds (dfh_gn_build_planned_assoc_views,
 * dfh_gn_iteration_views) drop, per 128-sample tile, the views none of its samples can be valid in -- the tile's warped samples'
 * box projects outside the image, or onto pixels whose valid depths all lie further than max_dist from the box's depth range
 * (exact for rigid extrinsics and a pinhole K: |c - x'| >= |z - l2| / scale) -- before projecting a single sample into them:
 * same corr / valid, same bits (option gn_no_view_cull = 1 keeps every view).  Round 4; no reference counterpart. */
size_t dfh_gn_views_bytes_cells(int n_views, int H, int W);
int dfh_gn_pack_views_cells(void *views_out, int n_views, const void *const *depth, int H, int W, const double *lw_cam, void *stream);
int dfh_gn_associate_views(const double *sample_pos, const int *nbr, const double *weights, int n_samples, int knn,
                           const double *node_dq, const double lw_dq[8], const void *views, int n_views, int depth_dtype, int H, int W,
                           const double K[9], const double Kinv[9], double scale, const double center[3], double half,
                           double max_dist, double *corr_out, unsigned char *valid_out, void *stream);

/* Gauss-Newton normal equations of 0.5*|computef|^2 in 6-DoF left twists (dq_a <- exp(xi_a) (x) dq_a):
 * vals (n_blocks x 36, block-sparse rows row_ptr/col with sorted columns) <- J^T J, rhs (6 n_nodes) <-
 * J^T r, cost_count[0] <- 0.5 |r|^2, cost_count[1] <- number of valid samples.  Data rows use the static
 * weights of dfh_sample_knn; node_nbr == NULL or rw == 0 skips the regularisation rows.  The block
 * pattern must contain every node pair of every sample tuple and every (i, j) of node_nbr (both
 * orders) plus the diagonal; missing blocks are silently dropped. */
int dfh_gn_build(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                 const double *corr, const unsigned char *valid, int n_samples, int knn, const double *node_dq,
                 const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                 const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                 double *rhs, double *cost_count, void *stream);

/* The same normal equations without floating-point atomics in the data term (same bits every run), for
 * callers that prepare a plan once per frame (samples and their node tuples are static while the warp field
 * moves).  Samples sorted by node tuple; a "row" = a maximal run of equal tuples inside one tile of dfh_gn_tile_samples() (128) samples:
 *   run_id[s]           row of sample s (n_rows rows);  partial: scratch of n_rows x dfh_gn_partial_doubles(knn)
 *                       + 2 x ceil(n_samples / dfh_gn_tile_samples()) + n_rows doubles (the rows | {cost, count} per tile | one live flag per row)
 *   blk_ptr (n_blocks+1), blk_ent   for block b the entries row * knn^2 + sa * knn + sb (slots sa, sb of the
 *                                   row's tuple hold the block's row node and column node), any fixed order
 *   node_ptr (n_nodes+1), node_ent  for node a the entries row * knn + slot
 * The tile pass stores each row's {Gram matrix as whole 6x6 sub-blocks for the slot pairs sa <= sb, 36 contiguous doubles each |
 * J^T r | cost | count} (rows padded to whole 64-byte lines; rows without a valid sample in this iteration are flagged dead in the
 * dense flag array and skipped); a gather pass adds them per block, reading one contiguous sub-block per list entry.
 * partial_reg (n_nodes * knn rows of dfh_gn_partial_doubles(2) doubles) + rblk_ptr / rblk_ent / rnode_ptr / rnode_ent: the same for the
 * regulariser, a pair (i, node_nbr[i*knn+slot]) being a 2-node row (entries row * 4 + sa * 2 + sb, row * 2 + slot);
 * partial_reg == NULL keeps the regulariser on atomics.
 * huber_delta > 0: every data row and its residual are scaled by sqrt(min(1, huber_delta / |r|)), the IRLS form of the
 * Huber loss the reference's solver uses (least_squares(loss='huber'), core/fusion.py:389); cost_count[0] is then the
 * Huber objective sum rho(r), rho = r^2 / 2 up to huber_delta and huber_delta (|r| - huber_delta / 2) beyond, plus the
 * regulariser's 0.5 |rho|^2.  0 = plain least squares. */
size_t dfh_gn_partial_doubles(int knn);
int dfh_gn_build_planned(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                         const double *corr, const unsigned char *valid, int n_samples, int knn, const double *node_dq,
                         const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                         const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                         double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                         const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                         const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                         void *stream);

/* dfh_gn_associate and dfh_gn_build_planned in ONE launch sequence: the data-row kernel warps every sample once, associates it
 * against `depth` (float32, H x W) exactly as dfh_gn_associate does -- corr_out / valid_out receive the same values, bit for
 * bit -- and sends the valid ones straight on to their Jacobian rows; the normal equations are those of dfh_gn_build_planned on
 * that corr / valid.  One launch and one blend + warp per sample less per GN iteration. */
int dfh_gn_build_planned_assoc(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                               double *corr_out, unsigned char *valid_out, int n_samples, int knn, const double *node_dq,
                               const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                               const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                               double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                               const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                               const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                               const float *depth, int H, int W, const double K[9], const double Kinv[9], const double lw_cam[12],
                               double scale, const double center[3], double half, double max_dist, void *stream);

/* Samples per tile of the planned build (a scratch row = a run of equal node tuples inside one tile; callers size tile_off,
 * the per-tile {cost, count} pairs behind the scratch rows and the torch restatement of the plan with it). */
int dfh_gn_tile_samples(void);

/* ---- per-frame bookkeeping of the planned build, on the device ---------------------------------------------------------
 * dfh_gn_sort_samples: the four per-sample arrays in the order of their node tuples (lexicographic, stable: equal tuples
 *   keep their input order), key_out[i] = the i-th sorted tuple as a knn-digit number in base n_nodes, order_out[i] =
 *   input index of the i-th sorted sample.
 * dfh_gn_plan_count:   rows = maximal runs of equal tuples inside tiles of dfh_gn_tile_samples() samples of the SORTED nbr; tile_off
 *   (ceil(n_samples / dfh_gn_tile_samples()) + 1 ints) <- first row of every tile, *n_rows_out (device) <- number of rows.
 *   (*n_rows_out, like *uncovered_out below and dfh_surface_count's *total_out, is written by ONE plain store of the sequence's
 *   last writer: it may be device memory or pinned host memory, which the host can then watch instead of queueing a copy.)
 * dfh_gn_plan_build:   run_id (n_samples), row_first (n_rows: first sample of every row) and the CSR lists of
 *   dfh_gn_build_planned -- blk_ptr (n_blocks + 1) / blk_ent (n_rows * knn^2), node_ptr (n_nodes + 1) / node_ent (n_rows * knn),
 *   every list in ascending entry order; *uncovered_out (device int) <- 1 if some node pair of some row is not a block
 *   of the pattern (row_ptr / col), its entries are left out.  n_rows is the value dfh_gn_plan_count produced. */
size_t dfh_gn_sort_workspace_bytes(int n_samples);
int dfh_gn_sort_samples(const double *pos, const double *nrm, const int *nbr, const double *weights, int n_samples, int knn,
                        int n_nodes, double *pos_out, double *nrm_out, int *nbr_out, double *weights_out, long *key_out,
                        int *order_out, void *workspace, size_t workspace_bytes, void *stream);
int dfh_gn_plan_count(const int *nbr, int n_samples, int knn, int *tile_off, int *n_rows_out, void *stream);
size_t dfh_gn_plan_workspace_bytes(int n_rows, int knn);
int dfh_gn_plan_build(const int *nbr, int n_samples, int knn, int n_nodes, const int *tile_off, int n_rows, const int *row_ptr,
                      const int *col, int n_blocks, int *run_id, int *row_first, int *blk_ptr, int *blk_ent, int *node_ptr,
                      int *node_ent, int *uncovered_out, void *workspace, size_t workspace_bytes, void *stream);

/* Block-Jacobi preconditioned CG on (A + lm_abs I + lm_rel diag(A)) x = -rhs, `iters` iterations, no
 * host synchronisation.  The damping is written into vals' diagonal (vals is consumed). */
size_t dfh_pcg_workspace_bytes(int n_nodes, int iters);
int dfh_pcg_solve(const int *row_ptr, const int *col, double *vals, const double *rhs, int n_nodes, int iters,
                  double lm_abs, double lm_rel, double *x_out, void *workspace, size_t workspace_bytes, void *stream);

/* dfh_pcg_solve followed by dfh_apply_twist(node_dq, x_out, n_nodes, step) in the same launch where the persistent
 * kernel runs (each row's wave updates its own node): one GN iteration's solve + update. */
int dfh_pcg_solve_update(const int *row_ptr, const int *col, double *vals, const double *rhs, int n_nodes, int iters,
                         double lm_abs, double lm_rel, double *x_out, void *workspace, size_t workspace_bytes, double *node_dq,
                         double step, void *stream);

/* One whole Gauss-Newton iteration of a single-GPU solve: dfh_gn_build_planned_assoc followed by dfh_pcg_solve_update on the
 * system it produced (vals / rhs -> x_out, node_dq <- exp(step * x) node_dq) -- the same bits as the two calls.  Knowing both
 * halves, the library lets the clearing of the solve's workspace ride in the data-row launch (its last workgroups) instead of
 * being a 5 us fill between gather and solve.  Multi-GPU solves keep the two calls (the all-reduce of vals / rhs / cost_count
 * goes between them).  Reference: the body of least_squares' iteration for Fusion.computef, core/fusion.py:356-389. */
int dfh_gn_iteration(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                     double *corr_out, unsigned char *valid_out, int n_samples, int knn, double *node_dq,
                     const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                     const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                     double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                     const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                     const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                     const float *depth, int H, int W, const double K[9], const double Kinv[9], const double lw_cam[12],
                     double scale, const double center[3], double half, double max_dist,
                     int pcg_iters, double lm_abs, double lm_rel, double *x_out, void *pcg_workspace, size_t pcg_workspace_bytes,
                     double step, void *stream);

/* dfh_gn_build_planned_assoc / dfh_gn_iteration with the association of dfh_gn_associate_views (float32 depth maps): the same
 * arguments with (views, n_views) in place of (depth, lw_cam).  dfh_gn_iteration_views queues `n_iters` whole iterations back to
 * back (a frame's ten iterations in one call: nothing between them depends on the host; the same bits as n_iters calls).
 * blk_upper (n_upper pairs of ints; NULL / 0: none): the symmetry of the block pattern -- for every block with column >= row
 * {its index, the index of its mirror block (column, row), -1 on the diagonal}.  With it the gather walks only those blocks'
 * lists and stores every sum twice, the second time transposed: J^T J is symmetric and block (b, a)'s list is block (a, b)'s with
 * the slots swapped, so the result is the same bit for bit with half the block walks and half the reads. */
int dfh_gn_build_planned_assoc_views(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                               double *corr_out, unsigned char *valid_out, int n_samples, int knn, const double *node_dq,
                               const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                               const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                               double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                               const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                               const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                               const void *views, int n_views, int H, int W, const double K[9], const double Kinv[9],
                               double scale, const double center[3], double half, double max_dist, const int *blk_upper, int n_upper,
                               void *stream);
int dfh_gn_iteration_views(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                     double *corr_out, unsigned char *valid_out, int n_samples, int knn, double *node_dq,
                     const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                     const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                     double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                     const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                     const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                     const void *views, int n_views, int H, int W, const double K[9], const double Kinv[9],
                     double scale, const double center[3], double half, double max_dist,
                     int pcg_iters, double lm_abs, double lm_rel, double *x_out, void *pcg_workspace, size_t pcg_workspace_bytes,
                     double step, int n_iters, const int *blk_upper, int n_upper, void *stream);
/* A frame's whole solve behind one call: n_global rigid-mode steps (a build + dfh_gn_global_step(global_lm) each; global_scratch as
 * there, global_xi_out may be NULL), then the n_iters node iterations of dfh_gn_iteration_views -- the same launches in the same
 * order as the separate calls, hence the same bits.  pipeline.SlabFrame.step's solve on one GPU. */
int dfh_gn_frame_solve_views(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights,
                     double *corr_out, unsigned char *valid_out, int n_samples, int knn, double *node_dq,
                     const double *node_pos, const double *node_w, const int *node_nbr, int n_nodes,
                     const double lw_dq[8], double rw, const int *row_ptr, const int *col, int n_blocks, double *vals,
                     double *rhs, double *cost_count, const int *run_id, int n_rows, double *partial, const int *blk_ptr,
                     const int *blk_ent, const int *node_ptr, const int *node_ent, double *partial_reg,
                     const int *rblk_ptr, const int *rblk_ent, const int *rnode_ptr, const int *rnode_ent, double huber_delta,
                     const void *views, int n_views, int H, int W, const double K[9], const double Kinv[9],
                     double scale, const double center[3], double half, double max_dist,
                     int pcg_iters, double lm_abs, double lm_rel, double *x_out, void *pcg_workspace, size_t pcg_workspace_bytes,
                     double step, int n_iters, const int *blk_upper, int n_upper,
                     int n_global, double global_lm, double *global_xi_out, void *global_scratch, size_t global_scratch_bytes, void *stream);

/* Multi-GPU solve: what travels in the per-iteration all-reduce.  `system` = {J^T J blocks (n_blocks x 36) | J^T r (6 n_nodes) |
 * cost, count} as the builds write it; J^T J is symmetric, so only the blocks with col >= row are packed (then J^T r and
 * {cost, count}): (36 n_upper + 6 n_nodes + 2) doubles in `packed`, about 55 % of the system.  row_of[b] / col[b]: block b's
 * row and column node; src[b]: index among the packed blocks of the one that holds block b's data (its own, or its mirror's
 * when col < row: the unpack transposes it).  No reference counterpart (the reference has no collective). */
int dfh_gn_pack_upper(const double *system, const int *row_of, const int *col, const int *src, int n_blocks, int n_nodes, int n_upper,
                      double *packed, void *stream);
int dfh_gn_unpack_upper(double *system, const int *row_of, const int *col, const int *src, int n_blocks, int n_nodes, int n_upper,
                        const double *packed, void *stream);

/* The persistent PCG kernel (one launch per solve; its workgroups synchronise through grid-wide reductions) is used
 * when all its workgroups are co-resident: the occupancy query admits a workgroup per CU and the grid needs at most
 * one workgroup per CU.  That cannot be known when several PROCESSES time-share one GPU: such callers declare it with
 * dfh_pcg_set_mode(2) and every solve then takes the two-launches-per-iteration path (0 = auto, the default).
 * A barrier of the persistent kernel that does not complete within its spin bound (seconds) makes every workgroup
 * leave: x_out = NaN, node_dq untouched (the twist update of dfh_pcg_solve_update / dfh_gn_iteration* is all or nothing: the
 * workgroup that finishes last applies every row's step, and only if no barrier timed out and every x is finite -- after a
 * timed-out solve node_dq is what it was before that solve), and a per-device counter is bumped.  dfh_pcg_status() synchronises `stream`,
 * reads and clears that counter: DFH_OK, or DFH_E_TIMEOUT when a solve since the last call timed out
 * (*aborted_solves_out = how many; may be NULL).  Call it wherever the host synchronises anyway. */
int dfh_pcg_set_mode(int mode);
/* Which path a solve with n_nodes rows takes on the current device right now: 1 = the persistent single-reduction kernel,
 * 2 = two launches per iteration (the textbook recurrence; same iterates in exact arithmetic, not the same bits), < 0 = error.
 * Callers that compare runs bit for bit (N ranks against one GPU) compare runs of the same path. */
int dfh_pcg_path(int n_nodes);
int dfh_pcg_status(void *stream, long *aborted_solves_out);
/* The same answer for the solves that have COMPLETED, without touching the device when none of them timed out (the kernel
 * also sets a word of pinned host memory): for callers that have just synchronised for a reason of their own (a count
 * read back) and do not want a second device round trip per frame.  Falls back to dfh_pcg_status() when the word is set. */
int dfh_pcg_status_peek(void *stream, long *aborted_solves_out);

/* node_dq[a] <- exp(step * xi[a]) (x) node_dq[a]; exp = rotation exp(omega), translation v. */
int dfh_apply_twist(double *node_dq, const double *xi, int n_nodes, double step, void *stream);
/* node_dq[a] <- exp(factor * log(node_dq[a])), 0 <= factor <= 1: every node's rigid motion scaled towards the identity along its
 * own screw (a unit dual quaternion comes back; zero / non-finite entries are left alone).  The composed frame loop calls it once
 * per frame after the TSDF update (pipeline.SlabFrame.step(relax=...)): Fusion.updateTSDF moves the canonical surface most of
 * the way to the live one every frame (core/fusion.py:180-190: the live sample weighs wi ~ tens against a canonical weight that
 * starts at the view count), so what the field carried is largely in the volume afterwards -- without this decay nothing ever
 * pulls a node back and the field random-walks.  No reference counterpart. */
int dfh_relax_twists(double *node_dq, int n_nodes, double factor, void *stream);
/* The rigid mode of a built system (dfh_gn_build*: vals, rhs), solved on its own: all nodes share ONE twist xi --
 * (sum of all 6x6 blocks + lm_rel diag) xi = -(sum of all J^T r) -- which is applied to every node,
 * node_dq[a] <- exp(xi) (x) node_dq[a], and written to xi_out (6 doubles, may be NULL).  Block-Jacobi PCG truncated at ten
 * iterations hardly moves this mode (the regulariser does not penalise it, the preconditioner does not see it); the frame loop
 * takes two such steps, each behind a build, before its node iterations (pipeline.SlabFrame.step(global_iters=...)).  The
 * reference fits a global rigid motion first too (Fusion.solve, precompute_lw: core/fusion.py:356-365).  scratch: device
 * memory of dfh_gn_global_step_bytes() bytes, ZEROED by the caller once (the kernel leaves it ready for the next call); sums
 * are added in a fixed order: the same bits every run.  Restated in oracle/gn_np.global_step. */
/* The same rigid-mode step from a SUBSAMPLE of the data rows, without a built system (what the frame loop takes): every
 * `stride`-th 128-sample tile of the (sorted) samples is associated against the views' table and differentiated as in the builds
 * (same Huber weights); a sample's Jacobian for the shared twist is the sum of its knn node blocks; the regulariser is left out (a
 * common left twist only rotates its residuals).  n_steps steps, each three short launches (rows, the 29 sums -- 21 upper entries
 * of A_g, 6 of g_g, objective, valid count -- and solve + apply); xi_out (8 doubles, may be NULL): the last step's twist | its
 * objective | its valid-sample count.  scratch: dfh_gn_global_sampled_bytes(n_samples, stride) bytes.  Sums in a fixed order: the
 * same bits every run.  knn = 4.  sums_out != NULL (n_steps = 1): only the 29 sums of THIS rank's samples are produced (32 doubles)
 * -- the caller all-reduces them over ranks and calls dfh_gn_global_apply: every rank then applies the same twist.
 * Restated in oracle/gn_np.global_step_sampled. */
size_t dfh_gn_global_sampled_bytes(int n_samples, int stride);
int dfh_gn_global_sampled_views(const double *sample_pos, const double *sample_nrm, const int *nbr, const double *weights, int n_samples, int knn,
                                double *node_dq, int n_nodes, const double lw_dq[8], double huber_delta, const void *views, int n_views, int H, int W,
                                const double K[9], const double Kinv[9], double scale, const double center[3], double half, double max_dist,
                                int stride, double lm_rel, int n_steps, double *xi_out, double *sums_out, void *scratch, size_t scratch_bytes,
                                void *stream);
int dfh_gn_global_apply(const double *sums29, double lm_rel, int n_nodes, double *node_dq, double *xi_out, void *stream);
size_t dfh_gn_global_step_bytes(void);
int dfh_gn_global_step(const double *vals, int n_blocks, const double *rhs, int n_nodes, double lm_rel, double *node_dq, double *xi_out,
                       void *scratch, size_t scratch_bytes, void *stream);

/* ---- surface samples for the solve (stand-in for marching cubes, core/fusion.py:554-568) -----------
 * Every band voxel (w > 0, |T| < band; T in voxel units as fuseDepths stores it) whose TSDF gradient
 * (central differences inside the slab, one-sided at its faces) is non-zero yields one sample: position
 * = voxel centre - T * gradient / (|gradient| * max(|gradient|, 1)) (one Newton step, never longer than |T|; global index space, plane 0 of the buffer is global plane x0), normal n =
 * gradient / |gradient|.  Samples come out in voxel order, deterministically.
 *   dfh_surface_count : per-block counts + exclusive scan into `workspace`, *total_out (device) = count
 *   dfh_surface_emit  : writes min(total, capacity) samples (n x 3 fp64 each); uses the same workspace.  capacity < total:
 *                       an even subsample in voxel order (sample i is kept iff it is the first with slot floor(i * capacity /
 *                       total), and stored in that slot), not a prefix. */
size_t dfh_surface_workspace_bytes(const int res[3]);
int dfh_surface_count(const void *tsdf, const void *tsdf_w, int vol_dtype, const int res[3], double band, void *workspace,
                      size_t workspace_bytes, long *total_out, void *stream);
int dfh_surface_emit(const void *tsdf, const void *tsdf_w, int vol_dtype, const int res[3], int x0, double band,
                     const void *workspace, double *pos_out, double *nrm_out, long capacity, void *stream);

/* ---- marching cubes: `_vertices` / `_faces` / `_normals` from a TSDF volume --------------------------
 * Replaces measure.marching_cubes_lewiner(volume, level, step_size, allow_degenerate=False) as called at
 * core/fusion_dm.py:319-331,342 and core/fusion.py:554-568 (skimage 0.13.1, a third-party dependency
 * that is not vendored).  Vertices sit on the edges of the step-subsampled lattice at the linearly
 * interpolated crossing of `level` (array-index coordinates, fp32 like skimage's), unit normals point
 * down the gradient (central differences on the lattice), faces are wound with their right-hand normal
 * up the gradient, zero-area faces are dropped -- the conventions of the reference's own output
 * meshes/original.obj.  Triangle choice inside a cube and the output order are this library's
 * (deterministic: vertices by owning lattice point then axis, faces by cube then table order).
 *   dfh_mc_count : totals_out[0] = vertices, [1] = faces, [2] = z rows (tiles) that emit anything
 *                  (3 device longs); fills `workspace`
 *   dfh_mc_emit  : writes min(total, capacity) vertices (x3 fp32), normals (x3 fp32), values (max of the
 *                  edge's two samples, may be NULL) and faces (x3 int32); same workspace, after dfh_mc_count.
 *                  n_active_tiles = totals_out[2] launches only those tiles; < 0 visits every tile. */
size_t dfh_mc_workspace_bytes(const int res[3], int step);
int dfh_mc_count(const void *vol, int vol_dtype, const int res[3], int step, double level, void *workspace,
                 size_t workspace_bytes, long *totals_out, void *stream);
int dfh_mc_emit(const void *vol, int vol_dtype, const int res[3], int step, double level, void *workspace,
                size_t workspace_bytes, float *verts, float *normals, float *values, int *faces, long cap_verts, long cap_faces,
                long n_active_tiles, void *stream);
/* The reference's vertex order on top of dfh_mc_emit's output: skimage numbers vertices as its faces create
 * them and flips the face rows afterwards, i.e. ids increase with first use when rows are read right-to-left
 * (meshes/original.obj).  Renumbers accordingly (faces rewritten in place, vertex arrays copied to *_out in
 * the new order), drops vertices no face uses; *used_out (device) = vertices kept. */
size_t dfh_mc_reorder_workspace_bytes(long n_verts, long n_faces);
int dfh_mc_reorder(const float *verts_in, const float *normals_in, const float *values_in, int *faces, long n_verts, long n_faces,
                   float *verts_out, float *normals_out, float *values_out, long *used_out, void *workspace,
                   size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DFUSION_HIP_H */
