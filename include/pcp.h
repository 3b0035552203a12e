/*
 * pcp.h -- C ABI of libpcp.so: the registration hot path of PointcloudProcessing
 * (ring nearest-neighbour, ring normals, point-to-point / point-to-plane
 * least-squares tiles, hierarchical 8x8 sum, device-resident ICP loop,
 * voxelizer / voxel stager) as hand-written gfx950 HIP kernels.
 *
 * The library replaces the reference's GL compute runtime,
 * /root/reference/ComputeShader.py (class ComputeShader), behind the same
 * call protocol; each entry point cites the method it stands in for.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure;
 *     pcp_last_error() returns the message of the calling thread's last failure
 *   - pointers are HOST pointers unless the name ends in _dev
 *   - one pcp_ctx per device and per pair in flight; calls on one ctx must be
 *     serialised by the caller (same rule as the GL context it replaces,
 *     ComputeShader.py:26)
 *   - points are float32 rows of 4 (x, y, z, w); scan lines are uint32 rows of
 *     4 (begin, end_inclusive, 1, 1) exactly as ComputeShader uploads them
 *   - calls that return results to host pointers wait for them.  Calls that only take host input
 *     (pcp_set_target, pcp_upload_source, pcp_upload_source_soa; the b_xyzw argument of a call that
 *     fails after its upload) enqueue the copy and return: the caller's buffer must stay valid and
 *     unchanged until the next call on that context that waits (pcp_synchronize, pcp_icp_fetch,
 *     pcp_batch_synchronize, any call that returns results)
 *   - a context adopted by a batch (pcp_batch_create) may be destroyed before the batch: pcp_destroy
 *     then drains the batch's stream and takes the context out of the batch
 */
#ifndef PCP_H
#define PCP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pcp_ctx pcp_ctx;

#define PCP_MODE_POINT 0 /* LeastSquaresPoint.glsl, outlier threshold 0.1 */
#define PCP_MODE_PLANE 1 /* LeastSquaresPlane.glsl, outlier threshold 0.2 */

const char* pcp_last_error(void);
int pcp_version(void);
/* Number of HIP devices visible to the process (no context is created). */
int pcp_device_count(int* count);

/* ComputeShader.__init__ (:13-98) + prepareBuffers (:124-203): device buffers
 * for clouds of at most max_points points (<= 2^24) and max_scans scan lines (<= 2048). */
int pcp_create(int device, int max_points, int max_scans, pcp_ctx** out);
/* ComputeShader.cleanup (:966-998). */
int pcp_destroy(pcp_ctx* ctx);
/* Run all later work of ctx on an existing hipStream_t (e.g. a torch side
 * stream). NULL restores the context's own stream; the legacy default stream,
 * whose handle is NULL, can therefore not be adopted. */
int pcp_set_stream(pcp_ctx* ctx, void* hip_stream);
int pcp_synchronize(pcp_ctx* ctx);

/* Upload half of prepareLS / preparePointPlane / prepareNNS
 * (ComputeShader.py:491-515, :571-585, :731-744): target cloud, scan lines
 * (already padded or not, as the caller's method requires; n_scans is
 * lens_data.z), origin as uploaded (4 floats, [ox,oy,oz,ox] after np.resize),
 * ring hints reset to -1 (setInitialScanGuess :488-489).  Also precomputes the
 * unit-sphere projection of every target point (FunctionalBase.glsl:41-48). */
int pcp_set_target(pcp_ctx* ctx, const float* a_xyzw, int n_a,
                   const uint32_t* scan_lines, int n_scans, const float* origin4);
/* The same with the target cloud already in device memory (e.g. pcp_frame_padded_dev);
 * scan lines and origin stay host pointers. */
int pcp_set_target_dev(pcp_ctx* ctx, const float* a_xyzw_dev, int n_a,
                       const uint32_t* scan_lines, int n_scans, const float* origin4);
/* setInitialScanGuess (:488-489) alone. */
int pcp_reset_hints(pcp_ctx* ctx);

/* prepareDispatchNormals (:541-567), kernel NormalShader.glsl.  out_normals
 * may be NULL (read=False); normals stay resident for pcp_ls_plane. */
int pcp_normals(pcp_ctx* ctx, float* out_normals /* n_a x 4 */);

/* dispatchNNS (:747-759), kernel NearestNeighbourScan.glsl.  hint_out may be
 * NULL; it receives corr[:,3] (the ring each match was found on). */
int pcp_nn(pcp_ctx* ctx, const float* b_xyzw, int n_b,
           uint32_t* idx, float* dist2, float* hint_out);

/* pcp_nn / pcp_ls_point / pcp_ls_plane for a source cloud as the caller holds it -- layout PCP_PTS_F64_XYZ (n rows of 3
 * float64, what PointcloudIcpContainer hands to dispatchNNS / dispatchLS / dispatchPointPlane, ComputeShader.py:747, :517,
 * :587), PCP_PTS_F64_ROWS, PCP_PTS_F32_XYZ or PCP_PTS_F32_XYZW (constants below) -- without the caller's pad_points pass (:956-961): the
 * rows are cast and padded chunk by chunk into a page-locked buffer of the context, each chunk uploaded while the next
 * is converted.  pcp_nn_staged leaves its results in page-locked arrays of the context (corr_out of the reference,
 * :231-241, is backend-owned in the same way): *idx, *dist2 and *hint (hint may be NULL) point at n_b entries that
 * stay valid until the next pcp_nn_staged call on the context.  pcp_ls_staged: mode PCP_MODE_POINT / PCP_MODE_PLANE. */
int pcp_nn_staged(pcp_ctx* ctx, const void* pts, int layout, int n_b, const uint32_t** idx, const float** dist2,
                  const float** hint);
int pcp_ls_staged(pcp_ctx* ctx, const void* pts, int layout, int n_b, int mode, int debug, float* hb /* 64 */);

/* Device time (ms, HIP events on the context's stream) of the search kernels of the last pcp_nn call,
 * without its upload and read-back. */
int pcp_nn_kernel_ms(pcp_ctx* ctx, float* ms);

/* dispatchLS (:517-529) = LeastSquaresPoint.glsl + the ParalellHBSum tree;
 * hb receives the summed 8x8 tile (H = hb[:6,:6], b = hb[:6,6]).
 * debug != 0 selects identity correspondences (debug_info.x >= 1). */
int pcp_ls_point(pcp_ctx* ctx, const float* b_xyzw, int n_b, int debug, float* hb /* 64 */);
/* dispatchPointPlane (:587-608) = LeastSquaresPlane.glsl + tree sum.
 * Requires pcp_normals() after the last pcp_set_target(). */
int pcp_ls_plane(pcp_ctx* ctx, const float* b_xyzw, int n_b, float* hb /* 64 */);
/* The per-point tiles Hs[n_b][8][8] of the same kernels (getBufferSubdata of
 * ssbo_Hs, :399-431), materialised; for inspection and tests. */
int pcp_ls_tiles(pcp_ctx* ctx, const float* b_xyzw, int n_b, int mode, int debug,
                 float* tiles /* n_b x 64 */);

/* prepareDispatchConcurrentHbSum (:630-727) + readSummedHB (:611-628), kernel
 * ParalellHBSum.glsl: 64-ary tree sum of n 8x8 float tiles. */
int pcp_hb_sum(pcp_ctx* ctx, const float* tiles /* n x 64 */, int n, float* out /* 64 */);
/* Same on tiles already in device memory; scratch_dev must hold
 * (ceil(n/64) + ceil(n/4096)) x 64 floats.  Runs on the context's stream,
 * no host sync. */
int pcp_hb_sum_dev(pcp_ctx* ctx, const float* tiles_dev, int n, float* scratch_dev,
                   float* out_dev /* 64 */);
/* pcp_hb_sum_dev without a workspace argument and without any allocation: the tree's intermediate
 * levels live in a fixed array inside the library's code object (1.04 MB per GPU, held until the
 * process ends).  No counterpart in the reference; it is for callers that cannot allocate device
 * memory where they call from: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  n <= 262144 tiles (64 MB of tiles; the reference's buffers hold 200000), larger n fail
 * with a message.  Same kernels and same result, bit for bit, as pcp_hb_sum_dev.  Enqueues on the
 * context's stream, no host sync.  All calls on one GPU share that GPU's array: calls that overlap
 * on one GPU (different contexts or streams) race and must be serialised by the caller.  Nothing in
 * the array has to be cleared between calls: every level overwrites what the next one reads. */
int pcp_hb_sum_static(pcp_ctx* ctx, const float* tiles_dev, int n, float* out_dev /* 64 */);

/* Device-resident iteration loop of PointcloudIcpContainer.full_pt2pt (:71-100)
 * and full_pt2pl (:103-144): per iteration the fused NN + tile + tree-sum
 * kernels, the 6x6 solve (float64, rounded to float32 like np.linalg.solve on
 * float32 input, :27,:59), rotation() (PointcloudAlignment.py:229-261) and
 * apply_iteration_of_ls (:172-174, float64) all run on the device.
 *   b_xyz     n_b x 3 float64 source points (as the reference keeps them)
 *   mode      PCP_MODE_POINT: exactly max_iter iterations
 *             PCP_MODE_PLANE: stops when mean|dx| < tol unless full_iter
 *   dx_trace  max_iter x 6 float32, the solved steps (t, then rotation angles)
 *   hb_trace  max_iter x 64 float32 or NULL, the summed tiles per iteration
 *   iters     number of dispatches executed
 *   converged 1 when the last dispatched step met tol (it is then NOT applied)
 *   b_out     n_b x 3 float64 or NULL, the moved source points
 * Requires pcp_set_target (and pcp_normals for the plane mode). */
int pcp_icp_run(pcp_ctx* ctx, const double* b_xyz, int n_b, int mode, int max_iter,
                float tol, int full_iter, float* dx_trace, float* hb_trace,
                int* iters, int* converged, double* b_out);

/* The same loop split for callers that keep data in device memory: upload
 * once, launch without host synchronisation, fetch the small result later. */
int pcp_upload_source(pcp_ctx* ctx, const double* b_xyz, int n_b);
/* The same for a cloud held as three separate coordinate arrays of n_b float64 each -- what a
 * transposed (3, n_b) result such as (R @ points.T).T of LidarDataReader.py:66-67 is in memory: the
 * rows are uploaded as they lie and interleaved on the device, no host copy into (n_b, 3) order. */
int pcp_upload_source_soa(pcp_ctx* ctx, const double* x, const double* y, const double* z, int n_b);
/* Same from a buffer already in device memory (device-to-device copy on the
 * context's stream). */
int pcp_upload_source_dev(pcp_ctx* ctx, const double* b_xyz_dev, int n_b);
int pcp_icp_launch(pcp_ctx* ctx, int mode, int max_iter, float tol, int full_iter);
int pcp_icp_fetch(pcp_ctx* ctx, float* dx_trace, float* hb_trace, int* iters,
                  int* converged, double* b_out);
/* The pose the finished loop returns to its caller, PointcloudIcpContainer.full_pt2pt (:84-100) / full_pt2pl
 * (:118-140): the 10x10x10 tracking grid of getUniformGrid (:429-438) moved by every applied step (applyIcpStep
 * :320-322, float64) and uniform_optimal_icp (:333-347) between the moved and the original grid -- means, 3x3
 * cross-covariance, R_opt = U V^T of its singular value decomposition (no determinant fix), t_opt = mean' - R_opt
 * mean -- computed on the device behind the loop (one more kernel on the context's stream).  The caller applies
 * p <- R_opt p - t_opt (EnvironmentConstructor.py:101-102).  t_opt3: 3 float64, R_opt9: row-major 3x3 float64.
 * Waits for the loop.  Also valid per context after pcp_batch_icp_launch. */
int pcp_icp_fetch_pose(pcp_ctx* ctx, double* t_opt3, double* R_opt9);
/* The same extraction for steps the caller holds: dx_steps = n_steps rows of 6 float32 (t, then the three angles),
 * applied in order to the tracking grid (rotation() of PointcloudAlignment.py:229-261 per step), then
 * uniform_optimal_icp.  n_steps = 0 gives the identity. */
int pcp_pose_from_steps(pcp_ctx* ctx, const float* dx_steps, int n_steps, double* t_opt3, double* R_opt9);
/* Recompute unit vectors (+ normals when with_normals) and reset hints for the
 * target already resident in ctx: the kernel half of prepare*, no upload. */
int pcp_prepare_resident(pcp_ctx* ctx, int with_normals);

/* ---- voxel map (Voxelizer.py's device half) ------------------------------
 * Layouts are the reference's: voxel_index int32 rows of 4 (vx, vy, vz,
 * data_row) sorted lexicographically; voxel_data int32 rows of max_points
 * ([0] = count, [1..] = global point ids); voxel_stat float32 rows of 4
 * (prob, count + max prob, stage status, staged) indexed by data_row. */

/* bufferVoxelIndex (ComputeShader.py:837-840). */
int pcp_voxel_index_upload(pcp_ctx* ctx, const int32_t* voxel_index, int rows);
/* fullBufferVoxelData (:842-845): replaces the resident voxel data. */
int pcp_voxel_data_upload(pcp_ctx* ctx, const int32_t* voxel_data, int rows, int max_points);
/* The sub-range branch of prepareDispatchVoxelizer (:863-878): rows
 * [first_row, first_row + n_rows) of the resident voxel data. */
int pcp_voxel_data_upload_rows(pcp_ctx* ctx, const int32_t* rows_data, int first_row,
                               int n_rows, int max_points);
/* getFullVoxelData (:946-954): the first `rows` rows back to the host. */
int pcp_voxel_data_read(pcp_ctx* ctx, int rows, int max_points, int32_t* out);
/* The dispatch half of prepareDispatchVoxelizer (:881-944), kernel
 * VoxelShader.glsl: every point is appended to its voxel's row or, when its
 * voxel is not in the table, to the unknown list.  unknown receives
 * [count, ids..., 0 padding] (unknown_len >= n + 1 ints, as the caller pads
 * it); debug (n x 4 floats: vx, vy, vz, table position) may be NULL. */
int pcp_voxelize(pcp_ctx* ctx, const float* pts_xyzw, int n, float voxel_size, int stored,
                 int begin_index, int max_points, int32_t* unknown, int unknown_len,
                 float* debug);
/* The same for points given as n x 3 float32 (x, y, z): what pad_points (:956-961) would append, a
 * zero fourth component, is added on the device, so the caller needs no padded copy. */
int pcp_voxelize_xyz(pcp_ctx* ctx, const float* pts_xyz, int n, float voxel_size, int stored,
                     int begin_index, int max_points, int32_t* unknown, int unknown_len,
                     float* debug);
/* The same for a cloud held as three coordinate rows (xyz_rows = n x values, then n y, then n z, float32):
 * what the float32 cast of a transposed cloud such as (R @ points.T).T of LidarDataReader.py:66-67 is in
 * memory; the rows are interleaved on the device. */
int pcp_voxelize_rows(pcp_ctx* ctx, const float* xyz_rows, int n, float voxel_size, int stored,
                      int begin_index, int max_points, int32_t* unknown, int unknown_len,
                      float* debug);
/* The host half of Voxelizer.storeUnknownPoints (Voxelizer.py:170-306) on the device, for the
 * unknown points of the LAST pcp_voxelize call (its points and unknown list are still resident):
 * one new voxel per distinct coordinate, in lexicographic order; data row stored + rank holds the
 * count of all its points and the first max_points - 1 ids (ascending); the entries are merged
 * into the sorted table.  Table and data grow on the device by the reference's rule (rows double
 * while new_stored >= rows, :238-252).  new_stored = voxels now in the table, rows_capacity = rows
 * the device data now has.  Fails if a voxel of an unknown point already exists (:228-229). */
int pcp_voxel_insert_unknown(pcp_ctx* ctx, int max_points, int* new_stored, int* rows_capacity);
/* Forget the resident voxel map (table and data are freed). */
int pcp_voxel_reset(pcp_ctx* ctx);
/* Device table / data of at least `rows` rows (contents kept, new data rows zero). */
int pcp_voxel_reserve(pcp_ctx* ctx, int rows, int max_points);
/* The first `rows` table entries (vx, vy, vz, data row) / the counts of n_rows data rows. */
int pcp_voxel_index_read(pcp_ctx* ctx, int rows, int32_t* voxel_index);
int pcp_voxel_counts_read(pcp_ctx* ctx, int first_row, int n_rows, int32_t* counts);
/* prepareDispatchVoxelStager (:761-835), kernel VoxelStage.glsl, on the table
 * and voxel data already resident.  voxel_stat (stat_rows x 4) is uploaded,
 * updated and read back in place; counter16 receives the counter buffer
 * ([0] = rows staged by this call); stage_out (may be NULL) receives the
 * first counter16[0] staged rows of max_points ints. */
int pcp_voxel_stage(pcp_ctx* ctx, float* voxel_stat, int stat_rows, int stored, int max_points,
                    int filter_outliers, int stage_everything, int max_staging,
                    int32_t* counter16, int32_t* stage_out);
/* pcp_voxel_stage with the statistics kept on the device between calls instead of uploaded and read back every frame
 * (16 bytes per voxel each way in prepareDispatchVoxelStager, ComputeShader.py:777-778, :809-811): rows the device has
 * not seen yet start as (0.5, 0, 0, 0), the statistics of a voxel that does not exist (Voxelizer.initVoxels,
 * Voxelizer.py:382-388); pcp_voxel_stage itself leaves what it uploaded resident, pcp_voxel_reset forgets everything.
 * pcp_voxel_stat_read copies the first `rows` rows (rows x 4 floats) to the host. */
int pcp_voxel_stage_resident(pcp_ctx* ctx, int stat_rows, int stored, int max_points, int filter_outliers,
                             int stage_everything, int max_staging, int32_t* counter16, int32_t* stage_out);
int pcp_voxel_stat_read(pcp_ctx* ctx, int rows, float* voxel_stat);
/* A device copy of the Voxelizer's global point store (added_points, filled by handleAddingPoints,
 * Voxelizer.py:398-425): while switched on, every pcp_voxelize* call also keeps its points at
 * [begin_index, begin_index + n) of a device array (16 bytes per point, doubling).  Switch it on before a map's
 * first dispatch; pcp_voxel_reset empties it, switching it off frees it. */
int pcp_voxel_store_enable(pcp_ctx* ctx, int on);
/* The points of the rows the last pcp_voxel_stage call staged, looked up in that store on the device: what the
 * reference gathers on the host per staged row before it hands the voxel on (Voxelizer.py:130-133 with
 * getStoredPointsFromVoxelDataSlice :396-399: ids = row[1 : min(row[0] + 1, max_points)]), row after row in
 * staging order as n_points x 3 float64 in device memory.  The buffer belongs to the context and is valid until
 * the next call of this function; the call returns when it is complete.  n_rows (or NULL) receives the rows staged. */
int pcp_voxel_staged_points(pcp_ctx* ctx, const double** xyz_dev, size_t* n_points, int* n_rows);

/* ---- a whole frame of Voxelizer.addPoints (Voxelizer.py:81-133) as one enqueue and one wait -----------------
 * What the reference does per frame in five round trips -- handleAddingPoints' float32 cast (:398-425), the
 * dispatch of VoxelShader.glsl (ComputeShader.prepareDispatchVoxelizer :847-944), the stager dispatch
 * (prepareDispatchVoxelStager :761-835, per-frame mode: stage_everything = False), the creation of the new voxels
 * (storeUnknownPoints :170-306) -- runs here as one stream of kernels on the device-resident map (pcp_voxel_reserve
 * / pcp_voxel_insert_unknown): every count the host used to read between the steps (unknown points, new voxels,
 * stored voxels, staged rows) stays in device memory, launches are sized for capacities the host can bound, and
 * capacity is reserved ahead of need, so the call only enqueues.  Results equal the step-by-step calls: table,
 * data rows (count of all points, the first max_points - 1 ids ascending), statistics, staged rows.
 *   layout   PCP_PTS_F64_XYZ: n rows of 3 float64 (cast to float32 as handleAddingPoints does);
 *            PCP_PTS_F32_XYZW / _XYZ / _ROWS: float32 rows of 4, rows of 3, or three coordinate rows of n
 *   stored   voxels in the table as the caller knows them after a step-by-step call, or -1: as the device knows them
 *            from the frames fed before (required after any step-by-step call that changes the table)
 *   serial   ticket for pcp_voxel_add_wait; at most 8 frames may be un-waited (the 9th call waits for the oldest)
 * pcp_voxel_add takes host memory (copied on the context's stream; the buffer must stay valid until the wait),
 * pcp_voxel_add_dev device memory produced on `after_stream` (NULL: already complete or on the context's stream).
 * The statistics are the resident ones of pcp_voxel_stage_resident.  A frame whose unknown points span more than
 * 2^16 voxels along an axis is refused at the wait (map unchanged by its unknown points). */
#define PCP_PTS_F32_XYZW 4
#define PCP_PTS_F32_XYZ 3
#define PCP_PTS_F32_ROWS (-3)
#define PCP_PTS_F64_XYZ 8
#define PCP_PTS_F64_ROWS (-8) /* three coordinate rows of n float64 (x row, y row, z row): a transposed (3, n) cloud such as
                                 (R @ points.T).T of LidarDataReader.py:66-67 as it lies in memory; pcp_nn_staged / pcp_ls_staged only */
typedef struct pcp_voxel_frame {
    int serial;          /* ticket */
    int stored_before;   /* voxels in the table when the frame arrived */
    int stored;          /* voxels in the table now */
    int unknown;         /* points whose voxel did not exist (unknown_point_inds[0]) */
    int created;         /* voxels created */
    int staged;          /* rows the stager released (counter_buffer[0][0]) */
    int key_bits;        /* bits of the voxel keys the creation sorted by */
    int flags;           /* bit 0: key range too wide, bit 1: a new voxel already existed, bit 2: coordinate beyond +-2^20 */
} pcp_voxel_frame;
int pcp_voxel_add(pcp_ctx* ctx, const void* pts_host, int layout, int n, float voxel_size, int stored, int begin_index,
                  int max_points, int filter_outliers, int max_staging, int* serial);
int pcp_voxel_add_dev(pcp_ctx* ctx, const void* pts_dev, int layout, int n, float voxel_size, int stored, int begin_index,
                      int max_points, int filter_outliers, int max_staging, void* after_stream, int* serial);
/* Waits for frame `serial` (and every frame before it) and returns its record; fails when the frame was refused. */
int pcp_voxel_add_wait(pcp_ctx* ctx, int serial, pcp_voxel_frame* out);
/* The unknown points of the last frame fed, on request (the reference returns them from every dispatch, :930-933):
 * ids[0 .. unknown) grouped by voxel in the table's order, ascending inside a voxel; ids == NULL returns the count
 * alone.  Waits. */
int pcp_voxel_add_unknown_read(pcp_ctx* ctx, int32_t* ids, int capacity, int* count);
/* The rows the last stager dispatch (either kind) released: rows x max_points ints, as prepareDispatchVoxelStager
 * returns them (:823-835).  Waits. */
int pcp_voxel_staged_read(pcp_ctx* ctx, int rows, int max_points, int32_t* stage_out);
/* Points [first, first + n) of the device point store (pcp_voxel_store_enable) as float32 rows of 4: what
 * Voxelizer.added_points holds for them (Voxelizer.py:398-425).  Waits. */
int pcp_voxel_store_read(pcp_ctx* ctx, long long first, int n, float* xyzw);
/* The hand-off of staged voxels (Voxelizer.addPoints :130-133: the points of every row the frame's stager released go to
 * the voxel manager) for frames that are fed without a wait in between: while a log is open, every frame of
 * pcp_voxel_add / _add_dev appends the points of the rows it staged -- looked up in the device point store as
 * pcp_voxel_staged_points does, row after row in staging order, n x 3 float64 -- to the log, in stream order, so a
 * producer that keeps many frames in flight (pcp_chain_submit) loses none of them.  Two logs (slot 0 and 1) let one run
 * of frames be taken while the next is written:
 *   _begin  empties the slot's log and opens it (capacity_points: the least room it must have; a slot's buffer only
 *           grows); the context's stream first waits for `after_stream` (or NULL), the stream that read the slot's
 *           previous contents;
 *   _end    closes it: frames fed afterwards are not logged; the counts start their way to the host;
 *   _take   waits for the frames fed while it was open and returns the log (device memory of the context, valid until
 *           the slot's next _begin), its points and the rows (voxels) they came from.  A log that overflowed fails here:
 *           the frame that did not fit left the log unchanged, its staged points are lost.
 * One log is open at a time; the step-by-step stager calls do not write to it.  Needs pcp_voxel_store_enable. */
int pcp_voxel_log_begin(pcp_ctx* ctx, int slot, size_t capacity_points, void* after_stream);
int pcp_voxel_log_end(pcp_ctx* ctx, int slot);
int pcp_voxel_log_take(pcp_ctx* ctx, int slot, const double** xyz_dev, size_t* n_points, int* n_rows);

/* ---- batches of independent frame pairs -----------------------------------
 * Consecutive frame pairs of a sequence are independent registrations
 * (EnvironmentConstructor.py:116-132 calls full_pt2pl once per frame).  A batch
 * runs the iteration loops of several prepared contexts of one device
 * together: per iteration ONE launch covers every pair (grid.y = pair) and one
 * more solves all pairs' 6x6 systems, so a 120k-point cloud no longer leaves
 * most of the 256 CUs idle.  Results per pair are those of pcp_icp_launch,
 * bit for bit.  No counterpart in the reference (it has no batching).
 * The contexts adopt the batch's stream until pcp_batch_destroy. */
typedef struct pcp_batch pcp_batch;
int pcp_batch_create(pcp_ctx** ctxs, int n, pcp_batch** out);
int pcp_batch_destroy(pcp_batch* batch);
/* pcp_prepare_resident for every context, on the batch's stream. */
int pcp_batch_prepare(pcp_batch* batch, int with_normals);
/* pcp_icp_launch for every context (each needs a target, a source, and normals
 * for the plane mode); fetch with pcp_icp_fetch per context. */
int pcp_batch_icp_launch(pcp_batch* batch, int mode, int max_iter, float tol, int full_iter);
int pcp_batch_synchronize(pcp_batch* batch);
/* As pcp_profile_enable / pcp_profile_read for the batched launches; points
 * receives the number of source points the timed launches processed. */
int pcp_batch_profile_enable(pcp_batch* batch, int on);
int pcp_batch_profile_read(pcp_batch* batch, float* ls_ms, int* launches, long long* points);

/* The per-ring elevation bands the search prunes with, n_scans x 4 floats:
 * (sin, cos) of the lowest and (sin, cos) of the highest unit-sphere elevation
 * among the ring's points; NaN rows mark rings that are never skipped.  For
 * inspection and tests; no counterpart in the reference. */
int pcp_ring_stats(pcp_ctx* ctx, float* out);

/* Per-kernel timing of the loop: while enabled, pcp_icp_launch records a HIP
 * event before and after every fused NN + least-squares launch on the
 * context's stream; pcp_profile_read waits for the stream and returns the
 * summed duration of the launches that did work and their count. */
int pcp_profile_enable(pcp_ctx* ctx, int on);
int pcp_profile_read(pcp_ctx* ctx, float* ls_ms, int* launches);

/* Elapsed milliseconds between two points of the context's stream, measured
 * with HIP events recorded on that stream. */
int pcp_timer_start(pcp_ctx* ctx);
int pcp_timer_stop(pcp_ctx* ctx, float* ms);

/* ---------------------------------------------------------------------------
 * Exact nearest neighbours on a cell grid (ring-less clouds, maps).
 * Stands in for the reference's scipy KD-tree: PointcloudIcpContainer.getKDTree
 * (:471-472), getNearestNeighbours (:467-469), icp_step_KD (:474-498) and
 * align_KD (:500-515).  Points are float64 rows of 3 (x, y, z), as the KD-tree
 * receives them.  One pcp_grid per device and map; calls on one grid must be
 * serialised by the caller.
 * ------------------------------------------------------------------------- */
typedef struct pcp_grid pcp_grid;

int pcp_grid_create(int device, pcp_grid** out);
int pcp_grid_destroy(pcp_grid* grid);
/* As pcp_set_stream. */
int pcp_grid_set_stream(pcp_grid* grid, void* hip_stream);
int pcp_grid_synchronize(pcp_grid* grid);
/* getKDTree (:471-472): index n map points with cubic cells of edge `cell`
 * (Morton-sorted, hash of occupied cells).  normals_xyzw (n rows of 4 float32, or
 * NULL) are carried along for the plane mode of pcp_grid_ls*. */
int pcp_grid_build(pcp_grid* grid, const double* xyz, size_t n, double cell, const float* normals_xyzw);
/* For a map split over ranks.  pcp_grid_partition cuts the Morton key space of partition cells (edge
 * own_cell) into `world` contiguous ranges with about equal numbers of map points: cuts receives
 * world + 1 keys, cuts[0] = 0, cuts[world] = 2^63, rank r owns [cuts[r], cuts[r+1]).
 * pcp_grid_build_slab is pcp_grid_build for the points whose partition cell, or any of the 26 cells
 * around it, has its key in [lo, hi) (one-cell halo), selected on the device; neighbour indices keep
 * referring to the whole array; n_kept = points held. */
int pcp_grid_partition(int device, const double* xyz, size_t n, double own_cell, int world, unsigned long long* cuts);
int pcp_grid_build_slab(pcp_grid* grid, const double* xyz, size_t n, double cell, const float* normals_xyzw,
                        double own_cell, unsigned long long lo, unsigned long long hi, size_t* n_kept);
/* New points join the resident map without a rebuild: the batch is sorted by cell key on its own, merged
 * into the map's order (a cell's points stay in the order of their ids) and the cell table is redone from
 * the merged keys.  This is how a map accumulates: the reference's Voxelizer hands the points of the voxels
 * it stages to its VoxelManager frame by frame (Voxelizer.py:130-133, VoxelManager.frameVoxelized :31-45);
 * here they go into the grid scans are registered against.  Ids are first_id + row index, so that feeding
 * batches with first_id = points fed before gives the map pcp_grid_build makes of all of them at once:
 * same order, same cells, same answers.  Needs a prior pcp_grid_build (which fixes the cell size; n = 0 is
 * fine).  A map with caller-supplied normals needs normals for every batch; normals estimated by
 * pcp_grid_estimate_normals are dropped by an append (estimate again).  The _slab form keeps only the
 * points in or next to the caller's share of the partition-cell key space, as pcp_grid_build_slab does:
 * every rank is handed the same batch and keeps its own part (n_kept). */
int pcp_grid_append(pcp_grid* grid, const double* xyz, size_t n, const float* normals_xyzw, long long first_id);
int pcp_grid_append_slab(pcp_grid* grid, const double* xyz, size_t n, const float* normals_xyzw, double own_cell,
                         unsigned long long lo, unsigned long long hi, long long first_id, size_t* n_kept);
/* pcp_grid_append / pcp_grid_append_slab for points already in device memory (n x 3 float64 on the grid's device,
 * e.g. what pcp_voxel_staged_points returns); no normals.  The work that produced the buffer must have completed
 * (the grid runs on a stream of its own); the buffer is read only and may be reused when the call returns. */
int pcp_grid_append_dev(pcp_grid* grid, const double* xyz_dev, size_t n, long long first_id);
int pcp_grid_append_slab_dev(pcp_grid* grid, const double* xyz_dev, size_t n, double own_cell, unsigned long long lo,
                             unsigned long long hi, long long first_id, size_t* n_kept);
/* Appends made between these two calls are collected -- coordinates and ids copied behind one another in device memory --
 * and join the map at pcp_grid_collect_end in ONE sort, merge and cell table instead of one per batch: for a producer that
 * hands over many small batches before anybody reads the map (the Voxelizer's final staging passes, Voxelizer.py:139-165:
 * a thousand voxels per dispatch).  The map is the one the same appends would have left one by one.  While appends are
 * being collected every call that reads or rebuilds the map fails; batches may not carry normals. */
int pcp_grid_collect_begin(pcp_grid* grid);
int pcp_grid_collect_end(pcp_grid* grid);
/* The same without a bracket to close: while on, every append (without normals) waits as if collected, and the first call
 * that reads the map -- a query, a registration, pcp_grid_info, pcp_grid_sorted_ids, normals -- takes everything that waits
 * in first, in one merge.  For a map that is fed every frame and read now and then (the Voxelizer's per-frame hand-off,
 * Voxelizer.py:130-133): an append then costs a copy, and the O(map) merge is paid once per reader.  A rebuild drops what
 * waits with the map it replaces; an append that carries normals takes the waiting batches in first; switching off flushes.
 * pcp_grid_collect_begin adopts what waits, and its pcp_grid_collect_end flushes as always. */
int pcp_grid_defer_appends(pcp_grid* grid, int on);
/* Facts about the last build: points, occupied cells, extent in cells, radix
 * passes, device time of the whole build and of the sort alone (ms). Any pointer may be NULL. */
int pcp_grid_info(pcp_grid* grid, unsigned long long* n_points, unsigned long long* n_cells, int* extent3,
                  int* sort_passes, float* build_ms, float* sort_ms);
/* The map in its sorted order: original index and coordinates of every entry. */
int pcp_grid_sorted_ids(pcp_grid* grid, long long* ids, double* xyz);
/* Normals for a map without scan rings.  The reference computes normals only along scan lines
 * (NormalShader.glsl:73-101), which a PLY cloud such as the ones align_KD (:500-515) registers does
 * not have; here the normal of a map point is the direction of least spread (smallest eigenvector of
 * the float64 scatter matrix) of the map points strictly within `radius` of it, itself included.
 * Points with fewer than max(min_points, 3) such points get a zero normal.  With viewpoint3 the
 * normal is turned towards that point (as normalTowardsOrigin, NormalShader.glsl:62-71), without it
 * its largest component is made positive.  radius may span at most 3 grid cells.  The normals stay
 * resident for the plane mode of pcp_grid_ls*; normals_xyzw (rows of 4 float32, w = 0) and counts
 * (points within the radius) receive them in the order of pcp_grid_sorted_ids; either may be NULL.
 * No counterpart in the reference. */
int pcp_grid_estimate_normals(pcp_grid* grid, double radius, int min_points, const double* viewpoint3,
                              float* normals_xyzw, unsigned* counts);
/* Upload m source points as the resident working copy (and a pristine copy that
 * pcp_grid_source_rewind restores). */
int pcp_grid_source(pcp_grid* grid, const double* xyz, size_t m);
int pcp_grid_source_rewind(pcp_grid* grid);
int pcp_grid_source_read(pcp_grid* grid, double* xyz);
/* applyIcpStep (:315-317): p <- R p + t on the working copy; minus_t != 0 gives
 * apply_iteration_of_ls (:172-174): p <- R p - t. */
int pcp_grid_source_apply(pcp_grid* grid, const double* R9, const double* t3, int minus_t);
/* KDTree.query (:468): exact nearest map point of every source point: idx = its
 * index in the array given to pcp_grid_build (-1 if none), dist = Euclidean
 * distance (+inf if none).  max_dist <= 0 or inf searches without bound, else
 * only points strictly nearer than max_dist count (distance_upper_bound).
 * Ties go to the lowest index.  kernel_ms (or NULL) receives the kernel's time. */
int pcp_grid_query_resident(pcp_grid* grid, double max_dist, long long* idx, double* dist, float* kernel_ms);
/* pcp_grid_source + pcp_grid_query_resident. */
int pcp_grid_query(pcp_grid* grid, const double* q_xyz, size_t m, double max_dist, long long* idx, double* dist);
/* icp_step_KD (:474-498) up to the SVD: nearest neighbours of the working copy,
 * the int(m * keep_fraction) pairs with the smallest distances (ties at the cut
 * by source index), their means and W = sum (ref - mean_ref)(other - mean_other)^T
 * (row-major 3x3, ref = map point).  The caller finishes with U, S, Vh = svd(W),
 * R = U Vh, t = mean_ref - R mean_other. */
int pcp_grid_kd_step(pcp_grid* grid, double keep_fraction, double* mean_ref3, double* mean_other3, double* W9,
                     unsigned long long* taken);
/* Per-point results of the last query / kd step / pcp_grid_ls: index, distance,
 * membership of the trimmed set (kd step only).  Any pointer may be NULL. */
int pcp_grid_nn_read(pcp_grid* grid, long long* idx, double* dist, unsigned char* selected);

/* Scan-to-map least squares.  The tile arithmetic is LeastSquaresPoint.glsl /
 * LeastSquaresPlane.glsl (thresholds 0.1 / 0.2 on the float32 squared distance)
 * and the ParalellHBSum tree over the source points in order, with the exact
 * neighbour within `radius` in place of the ring search; a source point without
 * such a neighbour contributes zeros.  No counterpart in the reference, which
 * registers scans against single frames only.
 * pcp_grid_ls: one dispatch on the working copy, H|b as dispatchLS returns them (8x8). */
int pcp_grid_ls(pcp_grid* grid, int mode, double radius, float* hb);
/* Device-resident loop on one GPU (the map analogue of pcp_icp_run): starts from
 * the working copy, leaves the registered cloud there. */
int pcp_grid_icp_run(pcp_grid* grid, int mode, double radius, int max_iter, float tol, int full_iter,
                     float* dx_trace, float* hb_trace, int* iters, int* converged);
/* The same loop cut at its one exchange point, for a map split over ranks: space is cut
 * into partition cells of edge own_cell (>= radius; independent of the grid's own cell),
 * each rank holds the map points in or next to its share of the partition-cell key space
 * and sums only the source points whose key k satisfies lo <= k < hi (keys: pcp_grid_cell_keys);
 * the partial 8x8 tiles are all-reduced by the caller between the two calls and every
 * rank solves the same system.  Both calls only enqueue work on the grid's stream;
 * tile_dev is a device buffer of 64 floats. */
int pcp_grid_set_owned_keys(pcp_grid* grid, double own_cell, unsigned long long lo, unsigned long long hi);
int pcp_grid_cell_keys(double cell, const double* xyz, size_t n, unsigned long long* keys);
int pcp_grid_icp_reset(pcp_grid* grid);
int pcp_grid_ls_dev(pcp_grid* grid, int mode, double radius, float* tile_dev);
int pcp_grid_pose_dev(pcp_grid* grid, const float* tile_dev, int mode, int max_iter, float tol, int full_iter);
/* Applies the step still pending after the last pcp_grid_pose_dev. */
int pcp_grid_icp_finish(pcp_grid* grid);
int pcp_grid_icp_fetch(pcp_grid* grid, float* dx_trace, float* hb_trace, int* iters, int* converged, int* done);

/* ---- the exchange step of a map split over GPUs, behind the ABI ---------------------------------
 * One rank per GPU (one process per GPU, or one thread per GPU of a process).  The communicator is
 * RCCL's (bound at run time: $PCP_RCCL_PATH, a copy already loaded into the process, or the loader
 * path); the only collective the path needs is the sum all-reduce of the 8x8 float32 tile between
 * pcp_grid_ls_dev and pcp_grid_pose_dev.  No counterpart in the reference (single GPU, no collective). */
typedef struct pcp_comm pcp_comm;
#define PCP_COMM_ID_BYTES 128
/* Rank 0 makes the id (ncclGetUniqueId) and hands its 128 bytes to the other ranks by any means. */
int pcp_comm_unique_id(char* id128);
/* Collective: every rank of the world calls it with the same id (ncclCommInitRank). */
int pcp_comm_init(int device, int world, int rank, const char* id128, pcp_comm** out);
int pcp_comm_destroy(pcp_comm* comm);
/* Tears the communicator down without waiting for collectives in flight (ncclCommAbort): what a rank calls after a
 * failure, so that its peers' pending all-reduces end with an error instead of waiting for it. */
int pcp_comm_abort(pcp_comm* comm);
int pcp_comm_info(pcp_comm* comm, int* world, int* rank);
/* In-place sum of 64 floats in device memory over all ranks, enqueued on hip_stream. */
int pcp_comm_allreduce_tile(pcp_comm* comm, float* tile_dev, void* hip_stream);
/* The same on the grid's stream, i.e. ordered between pcp_grid_ls_dev and pcp_grid_pose_dev. */
int pcp_grid_allreduce_tile(pcp_grid* grid, pcp_comm* comm, float* tile_dev);
/* pcp_grid_icp_run for a map split over the ranks of comm: every rank holds its share of the map
 * (pcp_grid_build_slab), its key range (pcp_grid_set_owned_keys) and the whole scan as working copy
 * (pcp_grid_source); per iteration partial tile -> all-reduce -> solve, every rank solving the same
 * system; all ranks return the same steps.  With a world of 1 the result equals pcp_grid_icp_run.  When the call fails
 * on one rank, that rank's caller must abort the communicator (pcp_comm_abort), which releases the peers still
 * waiting in an all-reduce with an error instead of leaving them there. */
int pcp_grid_icp_run_comm(pcp_grid* grid, pcp_comm* comm, int mode, double radius, int max_iter, float tol,
                          int full_iter, float* dx_trace, float* hb_trace, int* iters, int* converged);

/* ---------------------------------------------------------------------------
 * Frame preparation on the device: KITTI decode + ego filter + axis rotation
 * (LidarDataReader.eatBytes :144-150, :59-61, :66-67), the IMU pre-alignment
 * (EnvironmentConstructor.py:169, :209) and the ring segmentation
 * (PointcloudIcpContainer.getScanLines :362-392), so that a frame goes from the
 * file's bytes to the registration kernels without host-side array passes.
 * ------------------------------------------------------------------------- */
typedef struct pcp_frame pcp_frame;

int pcp_frame_create(int device, size_t max_points, pcp_frame** out);
int pcp_frame_destroy(pcp_frame* frame);
/* raw_xyzi: n float32 rows (x, y, z, intensity) as stored in a velodyne .bin.  Keeps the
 * rows with float32 |(x,y,z)| > min_range in file order, rotates them by R9 (row-major
 * float64 3x3) into the resident float64 cloud; n_out = rows kept. */
int pcp_frame_decode(pcp_frame* frame, const float* raw_xyzi, size_t n, float min_range, const double* R9,
                     size_t* n_out);
/* Make an existing float64 cloud the resident one. */
int pcp_frame_upload(pcp_frame* frame, const double* xyz, size_t n);
/* The same for a cloud held as three coordinate rows of n float64 each (a transposed (3, n) array such as
 * (R @ points.T).T of LidarDataReader.py:66-67 as it lies in memory); interleaved on the device. */
int pcp_frame_upload_rows(pcp_frame* frame, const double* x, const double* y, const double* z, size_t n);
/* p <- R p + t on the resident cloud. */
int pcp_frame_move(pcp_frame* frame, const double* R9, const double* t3);
/* getScanLines on the resident cloud: lines receives (begin, end_inclusive) pairs. */
int pcp_frame_scan_lines(pcp_frame* frame, const double* origin3, int max_lines, uint32_t* lines, int* n_lines);
int pcp_frame_read(pcp_frame* frame, double* xyz, float* intensity);
/* Device views of the resident cloud: float64 xyz (for pcp_upload_source_dev) and its
 * float32 padded copy (pad_points + cast, ComputeShader.py:956-961, :301-316; for
 * pcp_set_target_dev).  Valid until the next call that changes the frame. */
int pcp_frame_points_dev(pcp_frame* frame, const double** xyz_dev, size_t* n);
int pcp_frame_padded_dev(pcp_frame* frame, const float** xyzw_dev, size_t* n);

/* ---------------------------------------------------------------------------
 * A run of consecutive raw frames registered pair by pair as one stream of device work.
 * Per frame the reference reads the velodyne file (LidarDataReader.eatBytes :144-150), filters and
 * rotates it (:59-67), and per pair runs getScanLines (PointcloudIcpContainer.py:362-392), the
 * prepare* uploads and normals (ComputeShader.py:491-515, :571-585) and the iteration loop with one
 * read-back per iteration (full_pt2pt :71-100 / full_pt2pl :103-144, called once per frame by
 * EnvironmentConstructor.getRefinedTransition :116-132).  Here frame k of a submission is the target of
 * pair k and the source of pair k - 1; everything from the raw bytes to the solved steps runs on the
 * device without a host wait in between, the upload of one submission overlapping the registration of
 * the one before (separate copy stream).  Per-pair results are those of pcp_frame_decode +
 * pcp_frame_scan_lines + pcp_set_target_dev + pcp_icp_launch, bit for bit.
 * ------------------------------------------------------------------------- */
typedef struct pcp_seq pcp_seq;

/* Buffers for submissions of up to max_frames frames (max_frames - 1 pairs) of up to max_points raw points. */
int pcp_seq_create(int device, int max_frames, int max_points, int max_scans, pcp_seq** out);
int pcp_seq_destroy(pcp_seq* seq);
/* Enqueue n_frames raw frames (raw_xyzi[f]: n_raw[f] float32 rows x, y, z, intensity, as a velodyne .bin
 * holds them) and the registration of every consecutive pair: ego filter |p| > min_range, rotation R9
 * (row-major float64), scan lines about origin3, then mode / max_iter (<= 32) / tol / full_iter as
 * pcp_icp_launch.  Returns at once with a ticket; at most two submissions may be in flight.  The raw
 * buffers must stay valid and unchanged until the ticket is collected; the copy only overlaps other
 * work when they are page-locked (pcp_host_alloc). */
int pcp_seq_submit(pcp_seq* seq, const float* const* raw_xyzi, const int* n_raw, int n_frames, float min_range,
                   const double* R9, const double* origin3, int mode, int max_iter, float tol, int full_iter,
                   int* ticket);
/* pcp_seq_submit with a placement per frame, what EnvironmentConstructor.calculateTransition_imu (:154-265) does to a
 * frame before it is registered: placements holds 12 float64 per frame, a rotation Q (row-major) and an offset g.
 * Frame f becomes Q_f (R9 p) -- the IMU attitude of :169, a second, separately rounded product like
 * (imu_rotation @ cloud.T).T -- for its scan lines, its normals and as the target of pair f; as the source of pair
 * f - 1 it is that cloud + g_f (the IMU's guess of the motion since frame f - 1, :200-209), added in float64 when the
 * first dispatch reads the frame.  The steps returned are those of full_pt2pl(target, placed source): the pose of
 * the placed source against the target, the guess not included.  placements == NULL is pcp_seq_submit. */
int pcp_seq_submit_placed(pcp_seq* seq, const float* const* raw_xyzi, const int* n_raw, int n_frames, float min_range,
                          const double* R9, const double* origin3, const double* placements, int mode, int max_iter,
                          float tol, int full_iter, int* ticket);
/* Wait for a submission and fetch its results, pair p = (frame p <- frame p + 1): dx_trace receives
 * dx_stride_iters x 6 floats per pair (the steps, zero beyond iters[p]), iters / converged one int per
 * pair, n_points (or NULL) the points each of the n_frames frames kept.  Fails (after filling the outputs)
 * if a pair could not run: more scan lines than max_scans (as ComputeShader.py:506-512), a target without
 * a scan line, an empty source, singular normal equations. */
int pcp_seq_collect(pcp_seq* seq, int ticket, float* dx_trace, int dx_stride_iters, int* iters, int* converged,
                    int* n_points);
/* pcp_seq_collect that also returns every pair's pose as full_pt2pl / full_pt2pt return it (see pcp_icp_fetch_pose),
 * computed on the device behind the loops and fetched with the steps in the same copy: t_opt receives 3, R_opt 9
 * (row-major) float64 per pair; any output pointer may be NULL. */
int pcp_seq_collect_poses(pcp_seq* seq, int ticket, double* t_opt, double* R_opt, float* dx_trace, int dx_stride_iters,
                          int* iters, int* converged, int* n_points);
/* As pcp_batch_profile_enable / pcp_batch_profile_read: HIP events on the compute stream around every fused
 * NN + least-squares launch of the submissions made while enabled; read (after collecting) the summed
 * duration of the last collected submission's launches that did work, their number and the source points
 * they processed. */
int pcp_seq_profile_enable(pcp_seq* seq, int on);
int pcp_seq_profile_read(pcp_seq* seq, float* ls_ms, int* launches, long long* points);
/* The same launches one by one, in the order of the iterations: ms_each[i] (i < capacity) receives the duration of the
 * launch of iteration i, launches their number.  Iteration 0 searches all rings (hint -1) and runs its own kernel. */
int pcp_seq_profile_read_launches(pcp_seq* seq, float* ms_each, int capacity, int* launches);
/* ---------------------------------------------------------------------------
 * The per-frame driver loop of a drive as one stream of device work.
 * EnvironmentConstructor.calculateTransition_imu (EnvironmentConstructor.py:154-265) handles a frame on the host: guess =
 * refined position of the previous frame + dead-reckoned step (:198-206), placed = (imu_rotation @ cloud.T).T + guess
 * (:209), full_pt2pl / full_pt2pt against the reference cloud about its origin (:223-230; the frame that closes a block
 * of reset_every frames runs closing_iterations full iterations, :216-220), refined = (R_opt @ placed.T).T - t_opt and
 * position = R_opt @ guess - t_opt (:233-235), voxelizer.addPoints(refined) (:245), and every reset_every-th frame the
 * refined cloud and position replace the reference cloud and origin (:250-254).  A pcp_chain keeps all of that on the
 * device across frames and submissions: positions, the prepared reference (ring table, unit vectors, normals) and its
 * origin never visit the host, the refined clouds go to the voxel feed (pcp_voxel_add_dev) of voxel_ctx on that
 * context's stream while the next frame is registered.  Frame 0 of a drive is not registered: turned by its IMU
 * attitude it is the first reference cloud about the origin (0, 0, 0) (:168-189).
 * ------------------------------------------------------------------------- */
typedef struct pcp_chain pcp_chain;

/* Buffers for submissions of up to max_frames frames of up to max_points raw points each. */
int pcp_chain_create(int device, int max_frames, int max_points, int max_scans, pcp_chain** out);
int pcp_chain_destroy(pcp_chain* chain);
/* Forget the drive: the next frame submitted is frame 0 of a new one (EnvironmentConstructor.init :45-60). */
int pcp_chain_reset(pcp_chain* chain);
/* Enqueue the next n_frames raw frames of the drive (raw_xyzi[f]: n_raw[f] float32 rows x, y, z, intensity; ego filter
 * |p| > min_range and rotation R9 as pcp_seq_submit).  imu holds 12 float64 per frame: the IMU rotation of the frame
 * (row-major, current_rotation of :164) and its dead-reckoned step (getDeltaVelocity :90-95; ignored for frame 0 of a
 * drive).  mode / iterations / tol as pcp_icp_launch for ordinary frames (<= 32 iterations).  voxel_ctx (or NULL: no map) receives every
 * frame's final cloud through pcp_voxel_add_dev with voxel_size / voxel_max_points / filter_outliers / max_staging;
 * voxel_stored and voxel_begin_index are that call's `stored` and `begin_index` for the first frame fed (the following
 * frames continue from the device's own count and from the points fed so far); first_voxel_serial (or NULL) receives
 * the first frame's ticket for pcp_voxel_add_wait, the others follow consecutively.  The call waits once, for the number of points each frame keeps
 * (returned in n_points, n_frames ints, may be NULL), then enqueues and returns a ticket; at most two submissions may
 * be in flight.  Raw buffers must stay valid until the ticket is collected.  A call that fails while it enqueues (the
 * voxel feed refuses a frame, a launch fails) drains its streams and forgets the drive as pcp_chain_reset does: the
 * frames since the last collected submission are lost, the next frame submitted starts a new drive. */
int pcp_chain_submit(pcp_chain* chain, const float* const* raw_xyzi, const int* n_raw, int n_frames, float min_range,
                     const double* R9, const double* imu, int mode, int iterations, float tol, int closing_iterations,
                     int reset_every, pcp_ctx* voxel_ctx, float voxel_size, int voxel_stored, int voxel_begin_index,
                     int voxel_max_points, int filter_outliers, int max_staging, int* ticket, int* n_points,
                     int* first_voxel_serial);
/* Wait for a submission: per frame the estimated position (guess, 3 float64), the refined position (3), the pose
 * full_pt2pl returned (t_opt 3, R_opt 9 row-major; identity for frame 0 of a drive), the steps (dx_stride_iters x 6
 * float32, zero beyond iters[f]), iterations and convergence flag; chain_ms (or NULL) the device time of the
 * submission's chain on its stream.  Any output may be NULL.  Fails (after filling the outputs) when a reference cloud
 * could not be prepared (more scan lines than max_scans, none) or a frame met singular normal equations; such a frame
 * stays where the IMU put it. */
int pcp_chain_collect(pcp_chain* chain, int ticket, double* guess, double* position, double* t_opt, double* R_opt,
                      float* dx_trace, int dx_stride_iters, int* iters, int* converged, float* chain_ms);
/* The final cloud of frame `frame` of a collected submission (n x 3 float64: what went to the voxel map), while its
 * buffers have not been reused by the submission after the next. */
int pcp_chain_read_refined(pcp_chain* chain, int ticket, int frame, double* xyz, int n);
/* Page-locked host memory for buffers handed to pcp_seq_submit (hipHostMalloc / hipHostFree). */
int pcp_host_alloc(size_t bytes, void** out);
int pcp_host_free(void* p);

#ifdef __cplusplus
}
#endif
#endif /* PCP_H */
