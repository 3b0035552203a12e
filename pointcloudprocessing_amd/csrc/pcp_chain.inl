// The reference's per-frame driver loop as ONE stream of device work, included into pcp_registration.hip.
//
// EnvironmentConstructor.calculateTransition_imu (EnvironmentConstructor.py:154-265) does per frame, on the host:
//   guess    = refined position of the previous frame + dead-reckoned step               (:198-206)
//   placed   = (imu_rotation @ cloud.T).T + guess                                        (:209)
//   t, R     = full_pt2pl(reference cloud, placed, reference origin)                     (:223-230)
//              10 iterations with the early exit; the frame that closes a block of
//              `reset_every` frames runs 15 full iterations                              (:216-220)
//   refined  = (R @ placed.T).T - t ; position = R @ guess - t                           (:233-235)
//   voxelizer.addPoints(refined)                                                         (:245)
//   every `reset_every`-th frame: reference cloud = refined, origin = position           (:250-254)
// Frame k needs frame k - 1's position, so the frames of a drive are a chain; here the chain stays on the
// device: the guess is formed from the position the previous frame's kernels left in device memory, the
// registration loop reads a target descriptor that lives in device memory (the ring table and the origin of a
// replaced reference cloud are made by kernels, the host never learns them), the refined cloud is written by
// one kernel and handed to the voxel feed (pcp_voxel_add_dev) on the voxel context's stream, which overlaps the
// registration of the next frame.  The host enqueues; it waits once per run for the point counts of the
// decoded frames (the voxel feed and the launch sizes need them) and once for the results.
//
// Arithmetic follows NumPy's on the reference's expressions: (3, 3) @ (3, n) products are one fused chain per
// row (fma(r2, z, fma(r1, y, r0 x)), the float64 GEMM kernel), the 3-vector product R @ guess pairs the second
// product first (fma(r2, z, fma(r0, x, r1 y)), the GEMV kernel), additions and subtractions are plain.

namespace {

struct ChainState {
    double position[3];     // refined position of the last frame (prev_poses[-1][1])
    double origin[3];       // sensor position of the reference cloud (prev_origin)
    double guess[3];        // estimated_position of the frame in work
    int broken;             // the reference cloud could not be prepared (flags of k_chain_describe), sticky
    int n_lines;            // ring ends found in the reference cloud
};

struct ChainRecord {        // what the host wants back per frame
    int iters, converged, singular, flags, n, n_lines, closing, applied;
    double guess[3], position[3], pose[12];     // pose: t_opt[3], R_opt[9]
    float dx[PCP_SEQ_MAX_ITER * 6];
};

__global__ void k_chain_begin(ChainState* __restrict__ cs) {
    for (int k = 0; k < 3; ++k) { cs->position[k] = 0.0; cs->origin[k] = 0.0; cs->guess[k] = 0.0; }
    cs->broken = 0; cs->n_lines = 0;
}

// A cloud becomes the reference: its float32 padded copy (ComputeShader.pad_points :956-961 + the cast of the upload)
// and every point's azimuth about the origin the chain holds (getScanLines, PointcloudIcpContainer.py:366-374).
__global__ void k_chain_target(const double* __restrict__ xyz, int n, const ChainState* __restrict__ cs, float4* __restrict__ a32,
                               double* __restrict__ az) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = xyz[3 * (size_t)i + 0], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    a32[i] = make_float4((float)x, (float)y, (float)z, 0.0f);
    az[i] = ring_azimuth(x, y, z, cs->origin[0], cs->origin[1], cs->origin[2]);
}

struct ChainWrap {          // ring boundary after point i (getScanLines :376-381)
    const double* az;
    __device__ bool operator()(size_t i) const { return fabs(az[i + 1] - az[i]) > 6.0; }
};

struct ChainRingEnd {       // keeps the first max_ends ring ends; the count tells when there were more
    unsigned* ends;
    unsigned max_ends;
    __device__ void operator()(size_t i, size_t slot) const {
        if (slot < max_ends) ends[slot] = (unsigned)i;
    }
};

// The scan-line table as preparePointPlane / prepareLS would upload it (ComputeShader.py:571-585 pads to max_scans rows
// of (1, 1), :491-515 does not) and the target half of the loop's descriptor, from the ring ends the device found and the
// origin it holds.  One wave.  flags: bit 0 = more ring ends than max_scans, bit 1 = no ring.
__global__ void __launch_bounds__(64)
k_chain_describe(const unsigned* __restrict__ found_ptr, const unsigned* __restrict__ ends, int n_a, int max_scans, int pad_lines,
                 int2* __restrict__ scan, LsArgs* __restrict__ ls, PrepArgs* __restrict__ prep, ChainState* __restrict__ cs) {
    const unsigned found = *found_ptr;
    int flag = 0;
    if (found > (unsigned)max_scans) flag |= 1;
    if (found == 0) flag |= 2;
    int Z = 0, Zs = 0;
    if (!flag) {
        const int rows = pad_lines ? max_scans : (int)found;
        for (int k = threadIdx.x; k < rows; k += 64) {
            if (k < (int)found) scan[k] = make_int2(k == 0 ? 0 : (int)ends[k - 1] + 1, (int)ends[k]);
            else scan[k] = make_int2(1, 1);
        }
        Z = rows;
        // trailing rows equal to the row before them are padding copies: the search stops at the first of them
        Zs = Z;
        if (pad_lines && (int)found < max_scans) {
            const int last_begin = found == 1 ? 0 : (int)ends[found - 2] + 1, last_end = (int)ends[found - 1];
            Zs = (last_begin == 1 && last_end == 1) ? (int)found : (int)found + 1;
        }
    }
    if (threadIdx.x != 0) return;
    // origin as uploaded: np.resize(origin, (4,)) repeats, [ox, oy, oz, ox] (ComputeShader.py:436-437)
    const float4 o = make_float4((float)cs->origin[0], (float)cs->origin[1], (float)cs->origin[2], (float)cs->origin[0]);
    ls->n_a = flag ? 0 : n_a; ls->Z = Z; ls->Zs = Zs; ls->o = o;
    prep->n_a = flag ? 0 : n_a; prep->Z = Z; prep->o = o;
    cs->broken = flag;
    cs->n_lines = (int)found;
}

// placed = cloud + guess with guess = position of the previous frame + the frame's dead-reckoned step
// (EnvironmentConstructor.py:205, :209), the ring hints of a fresh source (setInitialScanGuess, ComputeShader.py:488-489)
// and the loop's state.
__global__ void k_chain_place(const double* __restrict__ xyz, int n, const double* __restrict__ step3, ChainState* __restrict__ cs,
                              double* __restrict__ b_in, float* __restrict__ hint, LsArgs* __restrict__ ls, IcpState* __restrict__ st) {
    const double gx = cs->position[0] + step3[0], gy = cs->position[1] + step3[1], gz = cs->position[2] + step3[2];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        b_in[3 * (size_t)i + 0] = xyz[3 * (size_t)i + 0] + gx;
        b_in[3 * (size_t)i + 1] = xyz[3 * (size_t)i + 1] + gy;
        b_in[3 * (size_t)i + 2] = xyz[3 * (size_t)i + 2] + gz;
        hint[i] = -1.0f;
    }
    if (i == 0) {
        const int broken = cs->broken;
        cs->guess[0] = gx; cs->guess[1] = gy; cs->guess[2] = gz;
        ls->n_b = broken ? 0 : n;
        st->pending = 0; st->iter = 0; st->done = broken ? 1 : 0; st->converged = 0; st->singular = 0;
    }
}

// refined = (R_opt @ placed.T).T - t_opt (EnvironmentConstructor.applyOptimalTranslation :101-102), the frame's refined
// position R_opt @ guess - t_opt (:97-99) and its record.  A loop that met singular normal equations leaves the frame
// where the IMU put it (identity pose) and says so in the record.
__global__ void k_chain_refine(const double* __restrict__ b_in, int n, const double* __restrict__ pose, ChainState* __restrict__ cs,
                               const IcpState* __restrict__ st, const float* __restrict__ dx_trace, double* __restrict__ refined,
                               ChainRecord* __restrict__ rec, int closing) {
    double R[9], t[3];
    const bool usable = pose[0] == pose[0] && !cs->broken;
    for (int k = 0; k < 9; ++k) R[k] = usable ? pose[3 + k] : (k % 4 == 0 ? 1.0 : 0.0);
    for (int k = 0; k < 3; ++k) t[k] = usable ? pose[k] : 0.0;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double x = b_in[3 * (size_t)i + 0], y = b_in[3 * (size_t)i + 1], z = b_in[3 * (size_t)i + 2];
        refined[3 * (size_t)i + 0] = fma(R[2], z, fma(R[1], y, R[0] * x)) - t[0];
        refined[3 * (size_t)i + 1] = fma(R[5], z, fma(R[4], y, R[3] * x)) - t[1];
        refined[3 * (size_t)i + 2] = fma(R[8], z, fma(R[7], y, R[6] * x)) - t[2];
    }
    if (blockIdx.x != 0) return;
    const int iters = min(st->iter, PCP_SEQ_MAX_ITER);
    for (int k = threadIdx.x; k < PCP_SEQ_MAX_ITER * 6; k += blockDim.x) rec->dx[k] = (k < iters * 6) ? dx_trace[k] : 0.0f;
    if (threadIdx.x != 0) return;
    const double gx = cs->guess[0], gy = cs->guess[1], gz = cs->guess[2];
    double p[3];
    for (int r = 0; r < 3; ++r) p[r] = fma(R[3 * r + 2], gz, fma(R[3 * r + 0], gx, R[3 * r + 1] * gy)) - t[r];
    rec->iters = iters; rec->converged = st->converged; rec->singular = st->singular; rec->flags = cs->broken;
    rec->n = n; rec->n_lines = cs->n_lines; rec->closing = closing; rec->applied = usable ? (int)pose[13] : 0;
    for (int k = 0; k < 3; ++k) { rec->guess[k] = cs->guess[k]; rec->position[k] = p[k]; rec->pose[k] = t[k]; }
    for (int k = 0; k < 9; ++k) rec->pose[3 + k] = R[k];
    for (int k = 0; k < 3; ++k) cs->position[k] = p[k];
    if (closing)
        for (int k = 0; k < 3; ++k) cs->origin[k] = p[k];
}

// Frame 0 of a drive: nothing is registered, its cloud is the first reference (:168-189).
__global__ void k_chain_first_record(const ChainState* __restrict__ cs, int n, ChainRecord* __restrict__ rec) {
    for (int k = threadIdx.x; k < PCP_SEQ_MAX_ITER * 6; k += blockDim.x) rec->dx[k] = 0.0f;
    if (threadIdx.x != 0) return;
    rec->iters = 0; rec->converged = 0; rec->singular = 0; rec->flags = cs->broken; rec->n = n; rec->n_lines = cs->n_lines;
    rec->closing = 1; rec->applied = 0;
    for (int k = 0; k < 3; ++k) { rec->guess[k] = 0.0; rec->position[k] = 0.0; rec->pose[k] = 0.0; }
    for (int k = 0; k < 9; ++k) rec->pose[3 + k] = (k % 4 == 0) ? 1.0 : 0.0;
}

}  // namespace

struct pcp_chain {
    int device = 0, max_frames = 0, max_points = 0, max_scans = 0;
    hipStream_t copy = nullptr, decode = nullptr, compute = nullptr;
    hipEvent_t uploaded = nullptr, counted = nullptr, finished[2] = {nullptr, nullptr}, fed[2] = {nullptr, nullptr};
    hipEvent_t t0[2] = {nullptr, nullptr}, t1[2] = {nullptr, nullptr};      // around a run's chain on the compute stream
    pcp_ctx* ctx = nullptr;                       // target-side buffers, working copy, loop state of the registration
    // per result slot (two submissions may be in flight): frame buffers and records
    std::vector<float4*> raw[2];
    std::vector<double*> xyz[2], refined[2];
    float4** raw_dev[2] = {nullptr, nullptr}; double** xyz_dev[2] = {nullptr, nullptr};
    int* n_raw_dev[2] = {nullptr, nullptr}; int* n_raw_host[2] = {nullptr, nullptr};
    double* imu_dev[2] = {nullptr, nullptr}; double* imu_host[2] = {nullptr, nullptr};      // per frame 12 doubles: Q, step
    unsigned* keep_counts[2] = {nullptr, nullptr}; unsigned* n_pts_dev[2] = {nullptr, nullptr}; unsigned* n_pts_host[2] = {nullptr, nullptr};
    ChainRecord* rec_dev[2] = {nullptr, nullptr}; ChainRecord* rec_host[2] = {nullptr, nullptr};
    int tiles = 0, stride = 0;
    // the reference cloud's preparation
    double* az = nullptr; unsigned* ends = nullptr; unsigned* ring_counts = nullptr;
    LsArgs* ls_dev = nullptr; PrepArgs* prep_dev = nullptr; ChainState* state = nullptr;
    int mode = -1;                                // mode the resident reference was prepared for
    // the drive
    long long total_frames = 0;                   // frames fed since the last reset (total_frame_counter)
    int local_frames = 0;                         // frames since the reference cloud was replaced (local_frame_counter)
    int n_reference = 0;                          // points of the reference cloud
    // submissions
    int in_flight[2] = {0, 0}, ticket_of[2] = {-1, -1};
    bool fed_voxels[2] = {false, false};
    std::vector<const double*> cloud_of[2];       // per frame of the slot's submission: where its final cloud lies
    long long submitted = 0;
};

namespace {

// Makes `xyz` (n points, float64, in device memory) the reference cloud: float32 copy, rings about the origin the chain
// holds, unit vectors, ring bands, normals for the plane mode.  Everything on the compute stream.
int chain_retarget(pcp_chain* s, const double* xyz, int n, int mode) {
    pcp_ctx* c = s->ctx;
    hipStream_t q = s->compute;
    hipLaunchKernelGGL(k_chain_target, dim3(cdiv(n, 256)), dim3(256), 0, q, xyz, n, s->state, c->a, s->az);
    PCP_CHECK(hipGetLastError());
    const int nblk = compact_tiles((size_t)n - 1);
    compact_ordered(q, ChainWrap{s->az}, ChainRingEnd{s->ends, (unsigned)s->max_scans}, (size_t)n - 1, s->ring_counts);
    PCP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_chain_describe, dim3(1), dim3(64), 0, q, s->ring_counts + nblk, s->ends, n, s->max_scans,
                       mode == PCP_MODE_PLANE ? 1 : 0, c->scan, s->ls_dev, s->prep_dev, s->state);
    hipLaunchKernelGGL(k_unit_batch, dim3(cdiv(s->max_points, 256), 1), dim3(256), 0, q, s->prep_dev);
    hipLaunchKernelGGL(k_ring_stats_batch, dim3(s->max_scans, 1), dim3(64), 0, q, s->prep_dev);
    if (mode == PCP_MODE_PLANE) {
        const size_t sm = (size_t)s->max_scans * (sizeof(int2) + sizeof(float4));
        hipLaunchKernelGGL(k_normals_batch, dim3(cdiv(n, PCP_BLOCK), 1), dim3(PCP_BLOCK), sm, q, s->prep_dev);
    }
    PCP_CHECK(hipGetLastError());
    s->n_reference = n;
    s->mode = mode;
    return 0;
}

}  // namespace

extern "C" {

int pcp_chain_destroy(pcp_chain* s) {
    if (!s) return 0;
    (void)hipSetDevice(s->device);
    for (hipStream_t q : {s->copy, s->decode, s->compute})
        if (q) (void)hipStreamSynchronize(q);
    for (int k = 0; k < 2; ++k) {
        if (s->fed[k]) (void)hipEventSynchronize(s->fed[k]);
        for (float4* p : s->raw[k]) if (p) (void)hipFree(p);
        for (double* p : s->xyz[k]) if (p) (void)hipFree(p);
        for (double* p : s->refined[k]) if (p) (void)hipFree(p);
        void* dev[] = {s->raw_dev[k], s->xyz_dev[k], s->n_raw_dev[k], s->imu_dev[k], s->keep_counts[k], s->n_pts_dev[k], s->rec_dev[k]};
        for (void* p : dev) if (p) (void)hipFree(p);
        void* host[] = {s->n_raw_host[k], s->imu_host[k], s->n_pts_host[k], s->rec_host[k]};
        for (void* p : host) if (p) (void)hipHostFree(p);
        for (hipEvent_t ev : {s->finished[k], s->fed[k], s->t0[k], s->t1[k]}) if (ev) (void)hipEventDestroy(ev);
    }
    void* dev[] = {s->az, s->ends, s->ring_counts, s->ls_dev, s->prep_dev, s->state};
    for (void* p : dev) if (p) (void)hipFree(p);
    if (s->ctx) { s->ctx->stream = s->ctx->own_stream; pcp_destroy(s->ctx); }
    if (s->uploaded) (void)hipEventDestroy(s->uploaded);
    if (s->counted) (void)hipEventDestroy(s->counted);
    for (hipStream_t q : {s->copy, s->decode, s->compute})
        if (q) (void)hipStreamDestroy(q);
    delete s;
    return 0;
}

int pcp_chain_create(int device, int max_frames, int max_points, int max_scans, pcp_chain** out) {
    if (!out) return pcp_fail("null out pointer");
    *out = nullptr;
    if (max_frames < 1 || max_frames > 4096) return pcp_fail("max_frames = %d outside [1, 4096]", max_frames);
    if (max_points < 2 || max_points > (1 << 24)) return pcp_fail("max_points = %d outside [2, 2^24]", max_points);
    if (max_scans < 1 || max_scans > 2048) return pcp_fail("max_scans = %d outside [1, 2048]", max_scans);
    int ndev = 0;
    PCP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return pcp_fail("device %d not present (%d visible)", device, ndev);
    PCP_CHECK(hipSetDevice(device));
    pcp_chain* s = new pcp_chain();
    s->device = device; s->max_frames = max_frames; s->max_points = max_points; s->max_scans = max_scans;
    const int F = max_frames;
    s->tiles = compact_tiles((size_t)max_points);
    s->stride = s->tiles + 2;
    hipError_t e = hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking);
    auto ok = [&] { return e == hipSuccess; };
    if (ok()) e = hipStreamCreateWithFlags(&s->decode, hipStreamNonBlocking);
    if (ok()) e = hipStreamCreateWithFlags(&s->compute, hipStreamNonBlocking);
    if (ok()) e = hipEventCreateWithFlags(&s->uploaded, hipEventDisableTiming);
    if (ok()) e = hipEventCreateWithFlags(&s->counted, hipEventDisableTiming);
    auto A = [&](void** p, size_t bytes) { if (ok()) e = hipMalloc(p, bytes); };
    auto H = [&](void** p, size_t bytes) { if (ok()) e = hipHostMalloc(p, bytes); };
    for (int k = 0; k < 2; ++k) {
        if (ok()) e = hipEventCreateWithFlags(&s->finished[k], hipEventDisableTiming);
        if (ok()) e = hipEventCreateWithFlags(&s->fed[k], hipEventDisableTiming);
        if (ok()) e = hipEventCreate(&s->t0[k]);
        if (ok()) e = hipEventCreate(&s->t1[k]);
        s->raw[k].assign(F, nullptr); s->xyz[k].assign(F, nullptr); s->refined[k].assign(F, nullptr);
        for (int f = 0; f < F; ++f) {
            A((void**)&s->raw[k][f], (size_t)max_points * sizeof(float4));
            A((void**)&s->xyz[k][f], (size_t)max_points * 3 * sizeof(double));
            A((void**)&s->refined[k][f], (size_t)max_points * 3 * sizeof(double));
        }
        A((void**)&s->raw_dev[k], F * sizeof(void*)); A((void**)&s->xyz_dev[k], F * sizeof(void*));
        A((void**)&s->n_raw_dev[k], F * sizeof(int)); A((void**)&s->imu_dev[k], (size_t)F * 12 * sizeof(double));
        A((void**)&s->keep_counts[k], (size_t)F * s->stride * sizeof(unsigned)); A((void**)&s->n_pts_dev[k], F * sizeof(unsigned));
        A((void**)&s->rec_dev[k], (size_t)F * sizeof(ChainRecord));
        H((void**)&s->n_raw_host[k], F * sizeof(int)); H((void**)&s->imu_host[k], (size_t)F * 12 * sizeof(double));
        H((void**)&s->n_pts_host[k], F * sizeof(unsigned)); H((void**)&s->rec_host[k], (size_t)F * sizeof(ChainRecord));
    }
    A((void**)&s->az, (size_t)max_points * sizeof(double)); A((void**)&s->ends, (size_t)max_scans * sizeof(unsigned));
    A((void**)&s->ring_counts, compact_count_elems((size_t)max_points) * sizeof(unsigned));
    A((void**)&s->ls_dev, sizeof(LsArgs)); A((void**)&s->prep_dev, sizeof(PrepArgs)); A((void**)&s->state, sizeof(ChainState));
    if (!ok()) {
        pcp_fail("pcp_chain_create: %s", hipGetErrorString(e));
        pcp_chain_destroy(s);
        return 1;
    }
    if (pcp_create(device, max_points, max_scans, &s->ctx)) { pcp_chain_destroy(s); return 1; }     // message set by pcp_create
    pcp_ctx* c = s->ctx;
    c->stream = s->compute;
    // the pointer half of the descriptors never changes; sizes, ring counts and origin are written by kernels
    c->n_a = c->n_b = 0; c->n_scans = c->n_scans_search = 0;
    LsArgs ls = ls_args(c, 0, nullptr);
    PrepArgs prep;
    memset(&prep, 0, sizeof(prep));
    prep.a = c->a; prep.ua = c->ua; prep.normals = c->normals; prep.scan = c->scan; prep.stat = c->ring_stat;
    prep.bins = c->ring_bins; prep.hint = c->hint; prep.max_points = max_points;
    e = hipMemcpy(s->ls_dev, &ls, sizeof(LsArgs), hipMemcpyHostToDevice);
    if (ok()) e = hipMemcpy(s->prep_dev, &prep, sizeof(PrepArgs), hipMemcpyHostToDevice);
    for (int k = 0; k < 2 && ok(); ++k) {
        e = hipMemcpy(s->raw_dev[k], s->raw[k].data(), F * sizeof(void*), hipMemcpyHostToDevice);
        if (ok()) e = hipMemcpy(s->xyz_dev[k], s->xyz[k].data(), F * sizeof(void*), hipMemcpyHostToDevice);
    }
    if (!ok()) {
        pcp_fail("pcp_chain_create: %s", hipGetErrorString(e));
        pcp_chain_destroy(s);
        return 1;
    }
    *out = s;
    return 0;
}

int pcp_chain_reset(pcp_chain* s) {
    if (!s) return pcp_fail("null chain");
    if (s->in_flight[0] || s->in_flight[1]) return pcp_fail("collect the submissions in flight first");
    s->total_frames = 0;
    s->local_frames = 0;
    s->n_reference = 0;
    s->mode = -1;
    return 0;
}

int pcp_chain_submit(pcp_chain* s, const float* const* raw_xyzi, const int* n_raw, int n_frames, float min_range, const double* R9,
                     const double* imu, int mode, int iterations, float tol, int closing_iterations, int reset_every,
                     pcp_ctx* voxel_ctx, float voxel_size, int voxel_stored, int voxel_begin_index, int voxel_max_points,
                     int filter_outliers, int max_staging, int* ticket, int* n_points, int* first_voxel_serial) {
    if (!s) return pcp_fail("null chain");
    if (!raw_xyzi || !n_raw || !R9 || !imu || !ticket) return pcp_fail("null argument");
    if (n_frames < 1 || n_frames > s->max_frames) return pcp_fail("n_frames = %d outside [1, %d]", n_frames, s->max_frames);
    if (mode != PCP_MODE_POINT && mode != PCP_MODE_PLANE) return pcp_fail("unknown mode %d", mode);
    if (iterations < 1 || iterations > PCP_SEQ_MAX_ITER || closing_iterations < 1 || closing_iterations > PCP_SEQ_MAX_ITER)
        return pcp_fail("iterations = %d / %d outside [1, %d]", iterations, closing_iterations, PCP_SEQ_MAX_ITER);
    if (reset_every < 1) return pcp_fail("reset_every = %d: the reference cloud is replaced every >= 1 frames", reset_every);
    if (mode == PCP_MODE_PLANE && s->max_scans < 2) return pcp_fail("normals need at least 2 scan lines");
    if (s->total_frames > 0 && mode != s->mode) return pcp_fail("the drive's reference cloud was prepared for mode %d (pcp_chain_reset starts a new drive)", s->mode);
    if (voxel_ctx && voxel_ctx->device != s->device) return pcp_fail("the voxel context lives on device %d, the chain on %d", voxel_ctx->device, s->device);
    if (voxel_ctx && !(voxel_size > 0.0f)) return pcp_fail("voxel_size must be positive");
    for (int f = 0; f < n_frames; ++f) {
        if (!raw_xyzi[f]) return pcp_fail("frame %d: null points", f);
        if (n_raw[f] < 0 || n_raw[f] > s->max_points) return pcp_fail("frame %d: %d raw points outside [0, %d]", f, n_raw[f], s->max_points);
    }
    const int slot = (int)(s->submitted & 1);
    if (s->in_flight[slot]) return pcp_fail("two submissions are in flight: collect ticket %d first", s->ticket_of[slot]);
    PCP_CHECK(hipSetDevice(s->device));
    const int F = n_frames;
    s->cloud_of[slot].assign(F, nullptr);
    // ---- copy stream: raw frames and IMU rows (this slot's last user has been collected: its buffers are free)
    memcpy(s->n_raw_host[slot], n_raw, F * sizeof(int));
    memcpy(s->imu_host[slot], imu, (size_t)F * 12 * sizeof(double));
    PCP_CHECK(hipMemcpyAsync(s->n_raw_dev[slot], s->n_raw_host[slot], F * sizeof(int), hipMemcpyHostToDevice, s->copy));
    PCP_CHECK(hipMemcpyAsync(s->imu_dev[slot], s->imu_host[slot], (size_t)F * 12 * sizeof(double), hipMemcpyHostToDevice, s->copy));
    for (int f = 0; f < F; ++f)
        if (n_raw[f]) PCP_CHECK(hipMemcpyAsync(s->raw[slot][f], raw_xyzi[f], (size_t)n_raw[f] * sizeof(float4), hipMemcpyHostToDevice, s->copy));
    PCP_CHECK(hipEventRecord(s->uploaded, s->copy));
    // ---- decode stream: ego filter, axis rotation and IMU attitude of every frame as one ordered compaction; it runs
    // beside the chain of the submission before this one
    hipStream_t d = s->decode;
    PCP_CHECK(hipStreamWaitEvent(d, s->uploaded, 0));
    const SeqKeepFar keep{(const float4* const*)s->raw_dev[slot], s->n_raw_dev[slot], min_range};
    SeqDecoded dec;
    dec.raw = (const float4* const*)s->raw_dev[slot]; dec.xyz = s->xyz_dev[slot]; dec.a32 = nullptr; dec.place = s->imu_dev[slot];
    dec.az = nullptr; dec.ox = dec.oy = dec.oz = 0.0;
    memcpy(dec.R, R9, sizeof(dec.R));
    const dim3 tiles_f(s->tiles, F);
    hipLaunchKernelGGL(k_seq_count<SeqKeepFar>, tiles_f, dim3(FR_THREADS), 0, d, keep, s->keep_counts[slot], s->stride);
    hipLaunchKernelGGL(k_seq_scan, dim3(F), dim3(1024), 0, d, s->keep_counts[slot], s->stride, s->tiles);
    hipLaunchKernelGGL((k_seq_scatter<SeqKeepFar, SeqDecoded>), tiles_f, dim3(FR_THREADS), 0, d, keep, dec, s->keep_counts[slot], s->stride);
    hipLaunchKernelGGL(k_seq_kept, dim3(cdiv(F, 64)), dim3(64), 0, d, s->keep_counts[slot], s->stride, s->tiles, F, s->n_pts_dev[slot]);
    PCP_CHECK(hipGetLastError());
    PCP_CHECK(hipMemcpyAsync(s->n_pts_host[slot], s->n_pts_dev[slot], F * sizeof(unsigned), hipMemcpyDeviceToHost, d));
    PCP_CHECK(hipEventRecord(s->counted, d));
    // the one wait of the submission: launch sizes and the voxel feed need the number of points every frame kept
    PCP_CHECK(hipEventSynchronize(s->counted));
    const unsigned* kept = s->n_pts_host[slot];
    for (int f = 0; f < F; ++f) {
        if (n_points) n_points[f] = (int)kept[f];
        if (kept[f] < 2) return pcp_fail("frame %d keeps %u points beyond %g m: nothing to register", f, kept[f], (double)min_range);
    }
    // ---- compute stream: the chain
    hipStream_t q = s->compute;
    pcp_ctx* c = s->ctx;
    PCP_CHECK(hipStreamWaitEvent(q, s->counted, 0));
    PCP_CHECK(hipEventRecord(s->t0[slot], q));
    const size_t sm = (size_t)s->max_scans * (sizeof(int2) + sizeof(float4));
    long long vox_begin = voxel_begin_index;
    int vox_serial0 = -1;
    auto feed = [&](const double* cloud, int n) -> int {
        if (!voxel_ctx) return 0;
        int serial = -1;
        const int stored = (vox_serial0 < 0) ? voxel_stored : -1;
        if (int e = pcp_voxel_add_dev(voxel_ctx, cloud, PCP_PTS_F64_XYZ, n, voxel_size, stored, (int)vox_begin, voxel_max_points,
                                      filter_outliers, max_staging, q, &serial)) return e;
        if (vox_serial0 < 0) vox_serial0 = serial;
        vox_begin += n;
        return 0;
    };
    // A failure from here on leaves frames of this submission half enqueued: the streams are drained and the drive is
    // forgotten (as pcp_chain_reset does), so that nothing later builds on positions that were never reported.
    auto give_up = [&](int code) -> int {
        (void)hipStreamSynchronize(q);
        if (voxel_ctx) (void)hipStreamSynchronize(voxel_ctx->stream);
        s->total_frames = 0; s->local_frames = 0; s->n_reference = 0; s->mode = -1;
        pcp_append_error(" (the drive has been reset: its frames since the last collected submission are lost)");
        return code;
    };
    for (int f = 0; f < F; ++f) {
        const int n = (int)kept[f];
        ChainRecord* rec = s->rec_dev[slot] + f;
        if (s->total_frames == 0) {
            // the first frame of a drive: its cloud, turned by the IMU attitude, is the first reference (:168-189)
            hipLaunchKernelGGL(k_chain_begin, dim3(1), dim3(1), 0, q, s->state);
            if (int e = chain_retarget(s, s->xyz[slot][f], n, mode)) return give_up(e);
            hipLaunchKernelGGL(k_chain_first_record, dim3(1), dim3(64), 0, q, s->state, n, rec);
            PCP_CHECK(hipGetLastError());
            if (int e = feed(s->xyz[slot][f], n)) return give_up(e);
            s->cloud_of[slot][f] = s->xyz[slot][f];
            s->total_frames = 1;
            s->local_frames = 0;
            continue;
        }
        const bool closing = s->local_frames + 1 >= reset_every;          // :216-220
        const int iters = closing ? closing_iterations : iterations;
        const int full = closing ? 1 : 0;
        hipLaunchKernelGGL(k_chain_place, dim3(cdiv(n, 256)), dim3(256), 0, q, s->xyz[slot][f], n, s->imu_dev[slot] + 12 * (size_t)f + 9,
                           s->state, c->b_in, c->hint, s->ls_dev, c->state);
        // the all-rings search of the fresh source as a pre-pass with 8 lanes per query (measured on this chain: 4 / 8 / 16
        // lanes 0.473 / 0.448 / 0.463 ms per frame, none 0.559, one lane per query with pooled dense passes 0.528)
        hipLaunchKernelGGL(k_first_rings_ind<8>, dim3(cdiv(n * 8, PCP_BLOCK)), dim3(PCP_BLOCK), sm, q, s->ls_dev);
        c->n_b = n;
        const FinishArgs fin = finish_args(c);
        for (int it = 0; it < iters; ++it) {
            if (mode == PCP_MODE_POINT) hipLaunchKernelGGL(k_ls_quad_ind<PCP_MODE_POINT>, dim3(cdiv(n, 64)), dim3(256), sm, q, s->ls_dev);
            else hipLaunchKernelGGL(k_ls_quad_ind<PCP_MODE_PLANE>, dim3(cdiv(n, 64)), dim3(256), sm, q, s->ls_dev);
            hipLaunchKernelGGL(k_finish, dim3(1), dim3(1024), 0, q, fin, 1, mode, iters, tol, full);
        }
        hipLaunchKernelGGL(k_pose, dim3(1), dim3(PCP_POSE_THREADS), 0, q, fin);
        hipLaunchKernelGGL(k_chain_refine, dim3(cdiv(n, 256)), dim3(256), 0, q, c->b_in, n, c->pose, s->state, c->state, c->dx_trace,
                           s->refined[slot][f], rec, closing ? 1 : 0);
        PCP_CHECK(hipGetLastError());
        if (int e = feed(s->refined[slot][f], n)) return give_up(e);
        s->cloud_of[slot][f] = s->refined[slot][f];
        s->local_frames++;
        if (closing) {                                                      // :250-254
            if (int e = chain_retarget(s, s->refined[slot][f], n, mode)) return give_up(e);
            s->local_frames = 0;
        }
        s->total_frames++;
    }
    PCP_CHECK(hipMemcpyAsync(s->rec_host[slot], s->rec_dev[slot], (size_t)F * sizeof(ChainRecord), hipMemcpyDeviceToHost, q));
    PCP_CHECK(hipEventRecord(s->t1[slot], q));
    PCP_CHECK(hipEventRecord(s->finished[slot], q));
    s->fed_voxels[slot] = voxel_ctx != nullptr;
    if (voxel_ctx) PCP_CHECK(hipEventRecord(s->fed[slot], voxel_ctx->stream));      // the feed has read the refined clouds by then
    s->in_flight[slot] = F;
    s->ticket_of[slot] = (int)(s->submitted & 0x7fffffff);
    *ticket = s->ticket_of[slot];
    if (first_voxel_serial) *first_voxel_serial = vox_serial0;
    s->submitted++;
    return 0;
}

int pcp_chain_collect(pcp_chain* s, int ticket, double* guess, double* position, double* t_opt, double* R_opt, float* dx_trace,
                      int dx_stride_iters, int* iters, int* converged, float* chain_ms) {
    if (!s) return pcp_fail("null chain");
    const int slot = ticket & 1;
    if (ticket < 0 || !s->in_flight[slot] || s->ticket_of[slot] != ticket) return pcp_fail("ticket %d is not in flight", ticket);
    if (dx_trace && dx_stride_iters < 1) return pcp_fail("dx_trace needs room for at least one step per frame");
    PCP_CHECK(hipSetDevice(s->device));
    PCP_CHECK(hipEventSynchronize(s->finished[slot]));
    if (s->fed_voxels[slot]) PCP_CHECK(hipEventSynchronize(s->fed[slot]));
    const int F = s->in_flight[slot];
    s->in_flight[slot] = 0;
    if (chain_ms) PCP_CHECK(hipEventElapsedTime(chain_ms, s->t0[slot], s->t1[slot]));
    const ChainRecord* rec = s->rec_host[slot];
    int bad = -1;
    for (int f = 0; f < F; ++f) {
        const ChainRecord& r = rec[f];
        if (guess) memcpy(guess + 3 * (size_t)f, r.guess, 3 * sizeof(double));
        if (position) memcpy(position + 3 * (size_t)f, r.position, 3 * sizeof(double));
        if (t_opt) memcpy(t_opt + 3 * (size_t)f, r.pose, 3 * sizeof(double));
        if (R_opt) memcpy(R_opt + 9 * (size_t)f, r.pose + 3, 9 * sizeof(double));
        if (iters) iters[f] = r.iters;
        if (converged) converged[f] = r.converged;
        if (dx_trace) {
            const int keep = r.iters < dx_stride_iters ? r.iters : dx_stride_iters;
            memset(dx_trace + (size_t)f * dx_stride_iters * 6, 0, (size_t)dx_stride_iters * 6 * sizeof(float));
            memcpy(dx_trace + (size_t)f * dx_stride_iters * 6, r.dx, (size_t)keep * 6 * sizeof(float));
        }
        if ((r.flags || r.singular) && bad < 0) bad = f;
    }
    if (bad >= 0) {
        const ChainRecord& r = rec[bad];
        if (r.flags & 1) return pcp_fail("frame %d: the reference cloud's %d scan lines do not fit maxScans = %d", bad, r.n_lines, s->max_scans);
        if (r.flags & 2) return pcp_fail("frame %d: the reference cloud has no scan line", bad);
        return pcp_fail("frame %d: singular normal equations at iteration %d (the frame was left where the IMU put it)", bad, r.iters - 1);
    }
    return 0;
}

int pcp_chain_read_refined(pcp_chain* s, int ticket, int frame, double* xyz, int n) {
    if (!s) return pcp_fail("null chain");
    if (!xyz) return pcp_fail("null output");
    const int slot = ticket & 1;
    if (ticket < 0 || s->ticket_of[slot] != ticket) return pcp_fail("the buffers of ticket %d have been reused", ticket);
    if (s->in_flight[slot]) return pcp_fail("collect ticket %d first", ticket);
    if (frame < 0 || frame >= (int)s->cloud_of[slot].size()) return pcp_fail("frame %d outside the submission", frame);
    const ChainRecord& r = s->rec_host[slot][frame];
    if (n != r.n) return pcp_fail("frame %d holds %d points, the buffer %d", frame, r.n, n);
    PCP_CHECK(hipSetDevice(s->device));
    PCP_CHECK(hipMemcpy(xyz, s->cloud_of[slot][frame], (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
