// Frame-to-frame odometry of a run of consecutive raw frames as ONE stream of device work, included
// into pcp_registration.hip (it launches that file's batched kernels).
//
// What the reference does per frame on the host -- read the velodyne file (LidarDataReader.py:144-150),
// ego filter and axis rotation (:59-67), getScanLines (PointcloudIcpContainer.py:362-392), upload,
// normals, the iteration loop with a read-back per iteration (full_pt2pl :103-144, called once per
// frame by EnvironmentConstructor.getRefinedTransition :116-132) -- happens here for F frames = F - 1
// pairs (frame k is the target of pair k and the source of pair k - 1) without the host ever waiting:
//   copy stream     raw frames host -> device
//   compute stream  decode all frames | ring boundaries of the targets | scan-line tables and the
//                   per-pair descriptors, written BY THE DEVICE from the point and ring counts it
//                   found | unit vectors, ring bands, normals | the batched iteration loop |
//                   the small per-pair results gathered into one record array | device -> host
// Everything the kernels need to know about a frame's size travels in device memory, so the launch
// dimensions are those of the capacity and surplus blocks leave at once.  The copy stream only waits
// for the decode of the previous submission (the raw buffers are free again then), so the upload of
// run s + 1 overlaps the registration of run s.

#define PCP_SEQ_MAX_ITER 32      // iterations per pair a run may ask for (the reference uses 10, 15 and 20)

namespace {

struct SeqKeepFar {       // ego filter of frame f: np.linalg.norm(float32 rows) > min_range
    const float4* const* raw;
    const int* n_raw;
    float min_range;
    __device__ size_t count(int f) const { return (size_t)n_raw[f]; }
    __device__ bool operator()(int f, size_t i) const {
        const float4 p = raw[f][i];
        const float s = (p.x * p.x + p.y * p.y) + p.z * p.z;
        return sqrtf(s) > min_range;
    }
};

// Azimuth of a point as getScanLines sees it (PointcloudIcpContainer.py:366-374): sphereProjectPoints about
// origin + (0, 0.127, 0), atan2(z, x) of the unit vector mapped to [0, 2 pi).
__device__ __forceinline__ double ring_azimuth(const double x, const double y, const double z, const double ox, const double oy,
                                               const double oz) {
    const double vx = x - ox, vy = (y - oy) - 0.127, vz = z - oz;
    const double mag = sqrt((vx * vx + vy * vy) + vz * vz);
    double a = atan2(vz / mag, vx / mag);
    if (a < 0.0) a += 6.283185307179586;
    return a;
}

struct SeqDecoded {       // rotated float64 point of frame f; targets also get their float32 padded copy
    const float4* const* raw;
    double* const* xyz;
    float4* const* a32;           // a32[f] == nullptr for the last frame (it is nobody's target); nullptr: no float32 copies
    const double* place;          // per frame 12 doubles (rotation Q row-major, offset g), or nullptr
    double* const* az;            // per frame: azimuth of every kept point about the origin (getScanLines :366-374); nullptr: none
    double ox, oy, oz;
    double R[9];
    __device__ void operator()(int f, size_t i, size_t slot) const {
        const float4 p = raw[f][i];
        const double x = (double)p.x, y = (double)p.y, z = (double)p.z;
        // the arithmetic of pcp_frame_decode (one fused chain per row)
        double rx = fma(R[2], z, fma(R[1], y, R[0] * x));
        double ry = fma(R[5], z, fma(R[4], y, R[3] * x));
        double rz = fma(R[8], z, fma(R[7], y, R[6] * x));
        if (place) {              // (imu_rotation @ cloud.T).T, EnvironmentConstructor.py:169, :209: a second, separately rounded product
            const double* Q = place + 12 * (size_t)f;
            const double qx = fma(Q[2], rz, fma(Q[1], ry, Q[0] * rx));
            const double qy = fma(Q[5], rz, fma(Q[4], ry, Q[3] * rx));
            const double qz = fma(Q[8], rz, fma(Q[7], ry, Q[6] * rx));
            rx = qx; ry = qy; rz = qz;
        }
        double* out = xyz[f];
        out[3 * slot + 0] = rx; out[3 * slot + 1] = ry; out[3 * slot + 2] = rz;
        // sphereProjectPoints about origin + (0, 0.127, 0), azimuth atan2(z, x) mapped to [0, 2 pi): once per point here,
        // the ring-boundary passes below only compare neighbours
        if (az) az[f][slot] = ring_azimuth(rx, ry, rz, ox, oy, oz);
        // ComputeShader.pad_points (:956-961) + the float32 cast of the upload
        if (a32 && a32[f]) a32[f][slot] = make_float4((float)rx, (float)ry, (float)rz, 0.0f);
    }
};

struct SeqWrap {          // ring boundary after point i of frame f (getScanLines :366-381)
    const double* const* az;      // azimuths written by the decode
    const unsigned* n_pts;        // points kept per frame (device)
    __device__ size_t count(int f) const { const unsigned n = n_pts[f]; return n < 2 ? 0 : (size_t)n - 1; }
    __device__ bool operator()(int f, size_t i) const {
        const double* a = az[f];
        return fabs(a[i + 1] - a[i]) > 6.0;
    }
};

struct SeqRingEnd {       // keeps the first max_ends ring ends; the count tells when there were more
    unsigned* const* ends;
    unsigned max_ends;
    __device__ void operator()(int f, size_t i, size_t slot) const {
        if (slot < max_ends) ends[f][slot] = (unsigned)i;
    }
};

// The three launches of the ordered compaction (see pcp_compact.h) for a batch of frames:
// blockIdx.y = frame, counts[f * stride ...] the frame's tile counts, counts[f * stride + tiles] its total.
template <class Pred>
__global__ void __launch_bounds__(FR_THREADS) k_seq_count(const Pred pred, unsigned* __restrict__ counts, int stride) {
    __shared__ unsigned total;
    const int f = blockIdx.y;
    const size_t n = pred.count(f);
    const size_t base = (size_t)blockIdx.x * FR_TILE;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    unsigned mine = 0;
    if (base < n) {
#pragma unroll
        for (int k = 0; k < FR_ITEMS; ++k) {
            const size_t i = base + (size_t)k * FR_THREADS + threadIdx.x;
            if (i < n && pred(f, i)) ++mine;
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) mine += __shfl_down(mine, s, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0) counts[(size_t)f * stride + blockIdx.x] = total;
}

// Exclusive prefix of a frame's tile counts in place, the total behind them.  One block per frame.
__global__ void __launch_bounds__(1024) k_seq_scan(unsigned* __restrict__ counts, int stride, int tiles) {
    __shared__ unsigned s[1024];
    __shared__ unsigned run;
    unsigned* c = counts + (size_t)blockIdx.x * stride;
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    for (int base = 0; base < tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const unsigned v = (i < tiles) ? c[i] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned t = (threadIdx.x >= (unsigned)off) ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < tiles) c[i] = run + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) run += s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) c[tiles] = run;
}

template <class Pred, class Emit>
__global__ void __launch_bounds__(FR_THREADS)
k_seq_scatter(const Pred pred, const Emit emit, const unsigned* __restrict__ counts, int stride) {
    __shared__ unsigned wave_total[FR_THREADS / 64];
    __shared__ unsigned running;
    const int f = blockIdx.y;
    const size_t n = pred.count(f);
    const size_t base = (size_t)blockIdx.x * FR_TILE;
    if (base >= n) return;                                   // block-uniform
    if (threadIdx.x == 0) running = counts[(size_t)f * stride + blockIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = 0; k < FR_ITEMS; ++k) {
        const size_t i = base + (size_t)k * FR_THREADS + threadIdx.x;
        const bool keep = i < n && pred(f, i);
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_total[w] = (unsigned)__popcll(bal);
        __syncthreads();
        unsigned before = running, all = 0;
        for (int j = 0; j < FR_THREADS / 64; ++j) { if (j < w) before += wave_total[j]; all += wave_total[j]; }
        if (keep) emit(f, i, (size_t)before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull)));
        __syncthreads();
        if (threadIdx.x == 0) running += all;
        __syncthreads();
    }
}

// Per frame: the number of points kept, out of the decode's counts.
__global__ void k_seq_kept(const unsigned* __restrict__ counts, int stride, int tiles, int n_frames,
                           unsigned* __restrict__ n_pts) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_frames) n_pts[f] = counts[(size_t)f * stride + tiles];
}

// Per pair p (target = frame p, source = frame p + 1): the scan-line table as preparePointPlane /
// prepareLS would upload it (ComputeShader.py:571-585 pads to max_scans rows of (1, 1), :491-515 does
// not), the sizes in the pair's three descriptors, its loop state.  One wave per pair.  flags[p]: bit 0 = more ring ends than max_scans, bit 1 = target without a ring,
// bit 2 = empty source; such a pair is switched off (n_b = 0, done = 1) and reported by the collect call.
struct SeqDescribe {
    const unsigned* n_pts; const unsigned* ring_counts; int ring_stride, ring_tiles;
    unsigned* const* ends; int2* const* scan;
    LsArgs* ls; FinishArgs* fin; PrepArgs* prep; int* flags;
    int n_pairs, max_scans, pad_lines;
    int last_iter;       // the run's last dispatch: the points it moves are never read (poses come from the steps)
    const double* place; // per frame 12 doubles (rotation, offset), or nullptr
};

__global__ void __launch_bounds__(64) k_seq_describe(const SeqDescribe D) {
    const int p = blockIdx.x;             // one wave per pair: a row per lane, every load independent of the others
    if (p >= D.n_pairs) return;
    const unsigned n_a = D.n_pts[p], n_b = D.n_pts[p + 1];
    const unsigned found = D.ring_counts[(size_t)p * D.ring_stride + D.ring_tiles];
    int flag = 0;
    if (found > (unsigned)D.max_scans) flag |= 1;
    if (found == 0) flag |= 2;
    if (n_b == 0) flag |= 4;
    int Z = 0, Zs = 0;
    if (!flag) {
        int2* scan = D.scan[p];
        const unsigned* ends = D.ends[p];
        const int rows = D.pad_lines ? D.max_scans : (int)found;
        for (int k = threadIdx.x; k < rows; k += 64) {
            if (k < (int)found) scan[k] = make_int2(k == 0 ? 0 : (int)ends[k - 1] + 1, (int)ends[k]);
            else scan[k] = make_int2(1, 1);
        }
        Z = rows;
        // Trailing rows equal to the row before them are padding copies (see set_target_impl): the search stops at the
        // first of them.  Real rows begin at strictly increasing points, so only the (1, 1) padding repeats: the
        // first padding row stays unless the last real row is (1, 1) itself.
        Zs = Z;
        if (D.pad_lines && (int)found < D.max_scans) {
            const int last_begin = found == 1 ? 0 : (int)ends[found - 2] + 1, last_end = (int)ends[found - 1];
            Zs = (last_begin == 1 && last_end == 1) ? (int)found : (int)found + 1;
        }
    }
    if (threadIdx.x != 0) return;
    LsArgs& L = D.ls[p];
    L.n_a = flag ? 0 : (int)n_a; L.n_b = flag ? 0 : (int)n_b; L.Z = Z; L.Zs = Zs;
    L.skip_store_iter = D.last_iter;
    D.fin[p].n1 = flag ? 0 : (int)((n_b + 63) / 64);
    D.prep[p].n_a = flag ? 0 : (int)n_a; D.prep[p].Z = Z;
    D.flags[p] = flag;
    IcpState* st = L.st;
    st->pending = 0; st->iter = 0; st->done = flag ? 1 : 0; st->converged = 0; st->singular = 0;
    if (D.place && !flag) {
        // The source is the frame as decoded plus its offset g (the "+ estimated_position" of EnvironmentConstructor.py:209): left as
        // a pending step (identity, -g) that the first dispatch applies when it reads the frame, x - (-g) == x + g in
        // float64.  It is no iteration: the loop's own steps are recorded from dispatch 0 on as always.
        const double* g = D.place + 12 * (size_t)(p + 1) + 9;
        for (int k = 0; k < 9; ++k) st->R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        st->t[0] = -g[0]; st->t[1] = -g[1]; st->t[2] = -g[2];
        st->pending = 1;
    }
}

__global__ void k_seq_origin(LsArgs* __restrict__ ls, PrepArgs* __restrict__ prep, int n_pairs, float4 o) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n_pairs) { ls[p].o = o; prep[p].o = o; }
}

// What the host wants back per pair, gathered into one array so that a single copy fetches a whole run.
struct SeqRecord {
    int iters, converged, singular, flags, n_a, n_b, n_lines, pad;
    double pose[12];                 // t_opt[3], R_opt[9] (uniform_optimal_icp of the applied steps)
    float dx[PCP_SEQ_MAX_ITER * 6];
};

__global__ void k_seq_results(const LsArgs* __restrict__ ls, const FinishArgs* __restrict__ fin, const int* __restrict__ flags,
                              const unsigned* __restrict__ n_pts, const unsigned* __restrict__ ring_counts, int ring_stride,
                              int ring_tiles, int n_pairs, SeqRecord* __restrict__ out) {
    const int p = blockIdx.x;
    if (p >= n_pairs) return;
    const IcpState* st = ls[p].st;
    SeqRecord* r = out + p;
    const int it = min(st->iter, PCP_SEQ_MAX_ITER);
    if (threadIdx.x == 0) {
        r->iters = it; r->converged = st->converged; r->singular = st->singular; r->flags = flags[p];
        r->n_a = (int)n_pts[p]; r->n_b = (int)n_pts[p + 1];
        r->n_lines = (int)ring_counts[(size_t)p * ring_stride + ring_tiles]; r->pad = 0;
    }
    for (int k = threadIdx.x; k < PCP_SEQ_MAX_ITER * 6; k += blockDim.x) r->dx[k] = (k < it * 6) ? fin[p].dx_out[k] : 0.0f;
    if (threadIdx.x < 12) r->pose[threadIdx.x] = flags[p] ? 0.0 : fin[p].pose[threadIdx.x];
}

}  // namespace

struct pcp_seq {
    int device = 0, max_frames = 0, max_points = 0, max_scans = 0;
    hipStream_t copy = nullptr, compute = nullptr;
    hipEvent_t uploaded = nullptr, decoded = nullptr, finished[2] = {nullptr, nullptr};
    bool decoded_pending = false;
    std::vector<pcp_ctx*> ctxs;                 // one per pair: target-side buffers, working copy, loop state
    std::vector<float4*> raw;                   // per frame
    std::vector<double*> xyz;
    std::vector<double*> az;                    // per frame: azimuth of every kept point
    std::vector<unsigned*> ends;
    // device arrays of per-frame pointers / sizes
    float4** raw_dev = nullptr; double** xyz_dev = nullptr; double** az_dev = nullptr; float4** a32_dev = nullptr;
    unsigned** ends_dev = nullptr;
    int2** scan_dev = nullptr;
    int* n_raw_dev = nullptr; int* n_raw_host[2] = {nullptr, nullptr};     // pinned mirrors, one per result slot
    // per-frame placements (12 doubles): one device copy per result slot too -- the describe kernel reads it after the
    // point at which the next submission's uploads may start
    double* place_dev[2] = {nullptr, nullptr}; double* place_host[2] = {nullptr, nullptr};
    unsigned *n_pts = nullptr, *keep_counts = nullptr, *ring_counts = nullptr;
    int tiles = 0, stride = 0;
    LsArgs* ls_dev = nullptr; FinishArgs* fin_dev = nullptr; PrepArgs* prep_dev = nullptr; int* flags_dev = nullptr;
    SeqRecord* rec_dev = nullptr; SeqRecord* rec_host[2] = {nullptr, nullptr};
    int in_flight[2] = {0, 0};                  // pairs of the submission in each result slot, 0 = free
    int ticket_of[2] = {-1, -1}, iters[2] = {0, 0};
    long long submitted = 0;
    bool profile = false;                       // bracket every fused NN + least-squares launch with events
    std::vector<hipEvent_t> prof_events;
    int prof_launches = 0;                      // launches bracketed by the last submission
    int last_longest = 0;                       // most iterations any pair of the last collected run made
    long long last_points = 0;                  // source points its launches processed
};

extern "C" {

int pcp_seq_destroy(pcp_seq* s) {
    if (!s) return 0;
    (void)hipSetDevice(s->device);
    if (s->copy) (void)hipStreamSynchronize(s->copy);
    if (s->compute) (void)hipStreamSynchronize(s->compute);
    for (pcp_ctx* c : s->ctxs) { c->stream = c->own_stream; pcp_destroy(c); }
    for (float4* p : s->raw) if (p) (void)hipFree(p);
    for (double* p : s->xyz) if (p) (void)hipFree(p);
    for (double* p : s->az) if (p) (void)hipFree(p);
    for (unsigned* p : s->ends) if (p) (void)hipFree(p);
    void* dev[] = {s->raw_dev, s->xyz_dev, s->az_dev, s->a32_dev, s->ends_dev, s->scan_dev, s->n_raw_dev, s->place_dev[0], s->place_dev[1], s->n_pts, s->keep_counts,
                   s->ring_counts, s->ls_dev, s->fin_dev, s->prep_dev, s->flags_dev, s->rec_dev};
    for (void* p : dev) if (p) (void)hipFree(p);
    for (int k = 0; k < 2; ++k) {
        if (s->n_raw_host[k]) (void)hipHostFree(s->n_raw_host[k]);
        if (s->place_host[k]) (void)hipHostFree(s->place_host[k]);
        if (s->rec_host[k]) (void)hipHostFree(s->rec_host[k]);
        if (s->finished[k]) (void)hipEventDestroy(s->finished[k]);
    }
    for (hipEvent_t ev : s->prof_events) (void)hipEventDestroy(ev);
    if (s->uploaded) (void)hipEventDestroy(s->uploaded);
    if (s->decoded) (void)hipEventDestroy(s->decoded);
    if (s->copy) (void)hipStreamDestroy(s->copy);
    if (s->compute) (void)hipStreamDestroy(s->compute);
    delete s;
    return 0;
}

int pcp_seq_create(int device, int max_frames, int max_points, int max_scans, pcp_seq** out) {
    if (!out) return pcp_fail("null out pointer");
    *out = nullptr;
    if (max_frames < 2 || max_frames > 4096) return pcp_fail("max_frames = %d outside [2, 4096]", max_frames);
    if (max_points < 2 || max_points > (1 << 24)) return pcp_fail("max_points = %d outside [2, 2^24]", max_points);
    if (max_scans < 1 || max_scans > 2048) return pcp_fail("max_scans = %d outside [1, 2048]", max_scans);
    int ndev = 0;
    PCP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return pcp_fail("device %d not present (%d visible)", device, ndev);
    PCP_CHECK(hipSetDevice(device));
    pcp_seq* s = new pcp_seq();
    s->device = device; s->max_frames = max_frames; s->max_points = max_points; s->max_scans = max_scans;
    const int F = max_frames, P = max_frames - 1;
    s->tiles = compact_tiles((size_t)max_points);
    s->stride = s->tiles + 2;
    hipError_t e = hipStreamCreateWithFlags(&s->copy, hipStreamNonBlocking);
    auto ok = [&] { return e == hipSuccess; };
    if (ok()) e = hipStreamCreateWithFlags(&s->compute, hipStreamNonBlocking);
    if (ok()) e = hipEventCreateWithFlags(&s->uploaded, hipEventDisableTiming);
    if (ok()) e = hipEventCreateWithFlags(&s->decoded, hipEventDisableTiming);
    for (int k = 0; k < 2 && ok(); ++k) {
        e = hipEventCreateWithFlags(&s->finished[k], hipEventDisableTiming);
        if (ok()) e = hipHostMalloc((void**)&s->rec_host[k], (size_t)P * sizeof(SeqRecord));
        if (ok()) e = hipHostMalloc((void**)&s->n_raw_host[k], F * sizeof(int));
        if (ok()) e = hipHostMalloc((void**)&s->place_host[k], (size_t)F * 12 * sizeof(double));
    }
    auto A = [&](void** p, size_t bytes) { if (ok()) e = hipMalloc(p, bytes); };
    s->raw.assign(F, nullptr); s->xyz.assign(F, nullptr); s->az.assign(F, nullptr); s->ends.assign(P, nullptr);
    for (int f = 0; f < F; ++f) {
        A((void**)&s->raw[f], (size_t)max_points * sizeof(float4));
        A((void**)&s->xyz[f], (size_t)max_points * 3 * sizeof(double));
        A((void**)&s->az[f], (size_t)max_points * sizeof(double));
        if (f < P) A((void**)&s->ends[f], (size_t)max_scans * sizeof(unsigned));
    }
    A((void**)&s->raw_dev, F * sizeof(void*)); A((void**)&s->xyz_dev, F * sizeof(void*)); A((void**)&s->az_dev, F * sizeof(void*));
    A((void**)&s->a32_dev, F * sizeof(void*)); A((void**)&s->ends_dev, P * sizeof(void*)); A((void**)&s->scan_dev, P * sizeof(void*));
    A((void**)&s->n_raw_dev, F * sizeof(int)); A((void**)&s->n_pts, F * sizeof(unsigned));
    A((void**)&s->place_dev[0], (size_t)F * 12 * sizeof(double)); A((void**)&s->place_dev[1], (size_t)F * 12 * sizeof(double));
    A((void**)&s->keep_counts, (size_t)F * s->stride * sizeof(unsigned));
    A((void**)&s->ring_counts, (size_t)P * s->stride * sizeof(unsigned));
    A((void**)&s->ls_dev, P * sizeof(LsArgs)); A((void**)&s->fin_dev, P * sizeof(FinishArgs));
    A((void**)&s->prep_dev, P * sizeof(PrepArgs)); A((void**)&s->flags_dev, P * sizeof(int));
    A((void**)&s->rec_dev, P * sizeof(SeqRecord));
    if (!ok()) {
        pcp_fail("pcp_seq_create: %s", hipGetErrorString(e));
        pcp_seq_destroy(s);
        return 1;
    }
    for (int p = 0; p < P; ++p) {
        pcp_ctx* c = nullptr;
        if (pcp_create(device, max_points, max_scans, &c)) { pcp_seq_destroy(s); return 1; }   // message set by pcp_create
        c->stream = s->compute;
        s->ctxs.push_back(c);
    }
    // the pointer halves of the descriptors never change: written once here, the sizes by k_seq_describe
    std::vector<LsArgs> ls(P);
    std::vector<FinishArgs> fin(P);
    std::vector<PrepArgs> prep(P);
    std::vector<float4*> a32(F, nullptr);
    std::vector<int2*> scan(P);
    for (int p = 0; p < P; ++p) {
        pcp_ctx* c = s->ctxs[p];
        c->n_a = c->n_b = 0; c->n_scans = c->n_scans_search = 0;
        ls[p] = ls_args(c, 0, nullptr);
        ls[p].b_in = s->xyz[p + 1];                 // the source of pair p IS the decoded frame p + 1
        fin[p] = finish_args(c);
        memset(&prep[p], 0, sizeof(PrepArgs));
        prep[p].a = c->a; prep[p].ua = c->ua; prep[p].normals = c->normals; prep[p].scan = c->scan; prep[p].stat = c->ring_stat;
        prep[p].bins = c->ring_bins; prep[p].hint = c->hint; prep[p].max_points = max_points;
        a32[p] = c->a;
        scan[p] = c->scan;
    }
    e = hipMemcpy(s->ls_dev, ls.data(), P * sizeof(LsArgs), hipMemcpyHostToDevice);
    if (ok()) e = hipMemcpy(s->fin_dev, fin.data(), P * sizeof(FinishArgs), hipMemcpyHostToDevice);
    if (ok()) e = hipMemcpy(s->prep_dev, prep.data(), P * sizeof(PrepArgs), hipMemcpyHostToDevice);
    if (ok()) e = hipMemcpy(s->raw_dev, s->raw.data(), F * sizeof(void*), hipMemcpyHostToDevice);
    if (ok()) e = hipMemcpy(s->xyz_dev, s->xyz.data(), F * sizeof(void*), hipMemcpyHostToDevice);
    if (ok()) e = hipMemcpy(s->az_dev, s->az.data(), F * sizeof(void*), hipMemcpyHostToDevice);
    if (ok()) e = hipMemcpy(s->a32_dev, a32.data(), F * sizeof(void*), hipMemcpyHostToDevice);
    if (ok()) e = hipMemcpy(s->ends_dev, s->ends.data(), P * sizeof(void*), hipMemcpyHostToDevice);
    if (ok()) e = hipMemcpy(s->scan_dev, scan.data(), P * sizeof(void*), hipMemcpyHostToDevice);
    if (!ok()) {
        pcp_fail("pcp_seq_create: %s", hipGetErrorString(e));
        pcp_seq_destroy(s);
        return 1;
    }
    *out = s;
    return 0;
}

int pcp_seq_submit_placed(pcp_seq* s, const float* const* raw_xyzi, const int* n_raw, int n_frames, float min_range,
                          const double* R9, const double* origin3, const double* placements, int mode, int max_iter, float tol,
                          int full_iter, int* ticket) {
    if (!s) return pcp_fail("null sequence");
    if (!raw_xyzi || !n_raw || !R9 || !origin3 || !ticket) return pcp_fail("null argument");
    if (n_frames < 2 || n_frames > s->max_frames) return pcp_fail("n_frames = %d outside [2, %d]", n_frames, s->max_frames);
    if (mode != PCP_MODE_POINT && mode != PCP_MODE_PLANE) return pcp_fail("unknown mode %d", mode);
    if (max_iter < 1 || max_iter > PCP_SEQ_MAX_ITER) return pcp_fail("max_iter = %d outside [1, %d]", max_iter, PCP_SEQ_MAX_ITER);
    if (mode == PCP_MODE_PLANE && s->max_scans < 2) return pcp_fail("normals need at least 2 scan lines");
    for (int f = 0; f < n_frames; ++f) {
        if (!raw_xyzi[f]) return pcp_fail("frame %d: null points", f);
        if (n_raw[f] < 0 || n_raw[f] > s->max_points) return pcp_fail("frame %d: %d raw points outside [0, %d]", f, n_raw[f], s->max_points);
    }
    const int slot = (int)(s->submitted & 1);
    if (s->in_flight[slot]) return pcp_fail("two submissions are in flight: collect ticket %d first", s->ticket_of[slot]);
    PCP_CHECK(hipSetDevice(s->device));
    const int F = n_frames, P = n_frames - 1;
    // ---- copy stream: the raw frames, once the previous run's decode has released the buffers
    if (s->decoded_pending) PCP_CHECK(hipStreamWaitEvent(s->copy, s->decoded, 0));
    memcpy(s->n_raw_host[slot], n_raw, F * sizeof(int));      // this slot's last user has been collected: its copy is done
    PCP_CHECK(hipMemcpyAsync(s->n_raw_dev, s->n_raw_host[slot], F * sizeof(int), hipMemcpyHostToDevice, s->copy));
    if (placements) {
        memcpy(s->place_host[slot], placements, (size_t)F * 12 * sizeof(double));
        PCP_CHECK(hipMemcpyAsync(s->place_dev[slot], s->place_host[slot], (size_t)F * 12 * sizeof(double), hipMemcpyHostToDevice, s->copy));
    }
    const double* place = placements ? s->place_dev[slot] : nullptr;
    for (int f = 0; f < F; ++f)
        if (n_raw[f]) PCP_CHECK(hipMemcpyAsync(s->raw[f], raw_xyzi[f], (size_t)n_raw[f] * sizeof(float4), hipMemcpyHostToDevice, s->copy));
    PCP_CHECK(hipEventRecord(s->uploaded, s->copy));
    // ---- compute stream
    hipStream_t q = s->compute;
    PCP_CHECK(hipStreamWaitEvent(q, s->uploaded, 0));
    const SeqKeepFar keep{(const float4* const*)s->raw_dev, s->n_raw_dev, min_range};
    SeqDecoded dec;
    dec.raw = (const float4* const*)s->raw_dev; dec.xyz = s->xyz_dev; dec.a32 = s->a32_dev; dec.place = place;
    dec.az = s->az_dev; dec.ox = origin3[0]; dec.oy = origin3[1]; dec.oz = origin3[2];
    memcpy(dec.R, R9, sizeof(dec.R));
    const dim3 tiles_f(s->tiles, F), tiles_p(s->tiles, P);
    hipLaunchKernelGGL(k_seq_count<SeqKeepFar>, tiles_f, dim3(FR_THREADS), 0, q, keep, s->keep_counts, s->stride);
    hipLaunchKernelGGL(k_seq_scan, dim3(F), dim3(1024), 0, q, s->keep_counts, s->stride, s->tiles);
    hipLaunchKernelGGL((k_seq_scatter<SeqKeepFar, SeqDecoded>), tiles_f, dim3(FR_THREADS), 0, q, keep, dec, s->keep_counts, s->stride);
    hipLaunchKernelGGL(k_seq_kept, dim3(cdiv(F, 64)), dim3(64), 0, q, s->keep_counts, s->stride, s->tiles, F, s->n_pts);
    PCP_CHECK(hipGetLastError());
    PCP_CHECK(hipEventRecord(s->decoded, q));
    s->decoded_pending = true;
    const SeqWrap wrap{(const double* const*)s->az_dev, s->n_pts};
    const SeqRingEnd ring_end{s->ends_dev, (unsigned)s->max_scans};
    hipLaunchKernelGGL(k_seq_count<SeqWrap>, tiles_p, dim3(FR_THREADS), 0, q, wrap, s->ring_counts, s->stride);
    hipLaunchKernelGGL(k_seq_scan, dim3(P), dim3(1024), 0, q, s->ring_counts, s->stride, s->tiles);
    hipLaunchKernelGGL((k_seq_scatter<SeqWrap, SeqRingEnd>), tiles_p, dim3(FR_THREADS), 0, q, wrap, ring_end, s->ring_counts, s->stride);
    SeqDescribe D;
    D.n_pts = s->n_pts; D.ring_counts = s->ring_counts; D.ring_stride = s->stride; D.ring_tiles = s->tiles;
    D.ends = s->ends_dev; D.scan = s->scan_dev; D.ls = s->ls_dev; D.fin = s->fin_dev; D.prep = s->prep_dev; D.flags = s->flags_dev;
    D.n_pairs = P; D.max_scans = s->max_scans; D.pad_lines = (mode == PCP_MODE_PLANE) ? 1 : 0;
    D.last_iter = max_iter - 1;
    D.place = place;
    hipLaunchKernelGGL(k_seq_describe, dim3(P), dim3(64), 0, q, D);
    PCP_CHECK(hipGetLastError());
    // origin as uploaded: np.resize(origin, (4,)) repeats, [ox, oy, oz, ox] (ComputeShader.py:436-437); it sits in the
    // descriptors' value half, which the host owns: refreshed only when it changes
    const float4 o4 = make_float4((float)origin3[0], (float)origin3[1], (float)origin3[2], (float)origin3[0]);
    bool stale = false;
    for (int p = 0; p < P; ++p) {
        pcp_ctx* c = s->ctxs[p];
        if (memcmp(&c->origin, &o4, sizeof(float4)) != 0) {
            c->origin = o4;
            stale = true;
        }
    }
    // the value travels as a kernel argument (copied when the launch is made), not as an asynchronous copy out of host
    // memory that a later submission may overwrite
    if (stale) hipLaunchKernelGGL(k_seq_origin, dim3(cdiv(P, 64)), dim3(64), 0, q, s->ls_dev, s->prep_dev, P, o4);
    // the kernel half of preparePointPlane / prepareLS for all pairs
    const size_t sm = (size_t)s->max_scans * (sizeof(int2) + sizeof(float4));
    hipLaunchKernelGGL(k_unit_batch, dim3(cdiv(s->max_points, 256), P), dim3(256), 0, q, s->prep_dev);
    hipLaunchKernelGGL(k_ring_stats_batch, dim3(s->max_scans, P), dim3(64), 0, q, s->prep_dev);
    if (mode == PCP_MODE_PLANE)
        hipLaunchKernelGGL(k_normals_batch, dim3(cdiv(s->max_points, PCP_BLOCK), P), dim3(PCP_BLOCK), sm, q, s->prep_dev);
    PCP_CHECK(hipGetLastError());
    // the iteration loops of all pairs; the registered clouds are not needed, so the last step is not applied
    const dim3 grid(cdiv(s->max_points, PCP_BLOCK), P), block(PCP_BLOCK);
    s->prof_launches = 0;
    for (int it = 0; it < max_iter; ++it) {
        if (s->profile) {
            while ((int)s->prof_events.size() < 2 * (it + 1)) {
                hipEvent_t ev;
                PCP_CHECK(hipEventCreate(&ev));
                s->prof_events.push_back(ev);
            }
            PCP_CHECK(hipEventRecord(s->prof_events[2 * it], q));
        }
        if ((it == 0 && use_pool() >= 1) || use_pool() >= 2) {
            if (mode == PCP_MODE_POINT) hipLaunchKernelGGL((k_ls_batch_first<PCP_MODE_POINT>), grid, block, sm, q, s->ls_dev);
            else hipLaunchKernelGGL((k_ls_batch_first<PCP_MODE_PLANE>), grid, block, sm, q, s->ls_dev);
        } else if (mode == PCP_MODE_POINT) hipLaunchKernelGGL((k_ls_batch<PCP_MODE_POINT>), dim3(cdiv(s->max_points, PCP_LS_THREADS), P), dim3(PCP_LS_THREADS), sm, q, s->ls_dev);
        else hipLaunchKernelGGL((k_ls_batch<PCP_MODE_PLANE>), dim3(cdiv(s->max_points, PCP_LS_THREADS), P), dim3(PCP_LS_THREADS), sm, q, s->ls_dev);
        if (s->profile) {
            PCP_CHECK(hipEventRecord(s->prof_events[2 * it + 1], q));
            s->prof_launches = it + 1;
        }
        hipLaunchKernelGGL(k_finish_batch, dim3(P), dim3(1024), 0, q, s->fin_dev, mode, max_iter, tol, full_iter);
    }
    PCP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(k_pose_batch, dim3(P), dim3(PCP_POSE_THREADS), 0, q, s->fin_dev);       // every pair's (t_opt, R_opt)
    hipLaunchKernelGGL(k_seq_results, dim3(P), dim3(64), 0, q, s->ls_dev, s->fin_dev, s->flags_dev, s->n_pts, s->ring_counts,
                       s->stride, s->tiles, P, s->rec_dev);
    PCP_CHECK(hipGetLastError());
    PCP_CHECK(hipMemcpyAsync(s->rec_host[slot], s->rec_dev, (size_t)P * sizeof(SeqRecord), hipMemcpyDeviceToHost, q));
    PCP_CHECK(hipEventRecord(s->finished[slot], q));
    s->in_flight[slot] = P;
    s->iters[slot] = max_iter;
    s->ticket_of[slot] = (int)(s->submitted & 0x7fffffff);
    *ticket = s->ticket_of[slot];
    s->submitted++;
    return 0;
}

int pcp_seq_submit(pcp_seq* s, const float* const* raw_xyzi, const int* n_raw, int n_frames, float min_range,
                   const double* R9, const double* origin3, int mode, int max_iter, float tol, int full_iter, int* ticket) {
    return pcp_seq_submit_placed(s, raw_xyzi, n_raw, n_frames, min_range, R9, origin3, nullptr, mode, max_iter, tol, full_iter, ticket);
}

int pcp_seq_collect(pcp_seq* s, int ticket, float* dx_trace, int dx_stride_iters, int* iters, int* converged, int* n_points) {
    return pcp_seq_collect_poses(s, ticket, nullptr, nullptr, dx_trace, dx_stride_iters, iters, converged, n_points);
}

int pcp_seq_collect_poses(pcp_seq* s, int ticket, double* t_opt, double* R_opt, float* dx_trace, int dx_stride_iters, int* iters,
                          int* converged, int* n_points) {
    if (!s) return pcp_fail("null sequence");
    const int slot = ticket & 1;
    if (ticket < 0 || !s->in_flight[slot] || s->ticket_of[slot] != ticket) return pcp_fail("ticket %d is not in flight", ticket);
    if (dx_trace && dx_stride_iters < s->iters[slot])      // before the slot is given up: the caller can come again
        return pcp_fail("dx_trace holds %d steps per pair, the run made up to %d", dx_stride_iters, s->iters[slot]);
    PCP_CHECK(hipSetDevice(s->device));
    PCP_CHECK(hipEventSynchronize(s->finished[slot]));
    const int P = s->in_flight[slot];
    s->in_flight[slot] = 0;
    const SeqRecord* rec = s->rec_host[slot];
    int bad = -1;
    s->last_longest = 0;
    s->last_points = 0;
    for (int p = 0; p < P; ++p) {
        const SeqRecord& r = rec[p];
        if (!r.flags) {
            s->last_longest = r.iters > s->last_longest ? r.iters : s->last_longest;
            s->last_points += (long long)r.iters * r.n_b;
        }
        if (iters) iters[p] = r.iters;
        if (converged) converged[p] = r.converged;
        if (n_points) { n_points[p] = r.n_a; n_points[p + 1] = r.n_b; }
        if (dx_trace) memcpy(dx_trace + (size_t)p * dx_stride_iters * 6, r.dx, (size_t)s->iters[slot] * 6 * sizeof(float));
        if (t_opt) memcpy(t_opt + 3 * (size_t)p, r.pose, 3 * sizeof(double));
        if (R_opt) memcpy(R_opt + 9 * (size_t)p, r.pose + 3, 9 * sizeof(double));
        if ((r.flags || r.singular) && bad < 0) bad = p;
    }
    if (bad >= 0) {
        const SeqRecord& r = rec[bad];
        if (r.flags & 1) return pcp_fail("pair %d: %d scan lines do not fit maxScans = %d", bad, r.n_lines, s->max_scans);
        if (r.flags & 2) return pcp_fail("pair %d: the target frame (%d points) has no scan line", bad, r.n_a);
        if (r.flags & 4) return pcp_fail("pair %d: the source frame is empty", bad);
        return pcp_fail("pair %d: singular normal equations at iteration %d", bad, r.iters - 1);
    }
    return 0;
}

int pcp_seq_profile_enable(pcp_seq* s, int on) {
    if (!s) return pcp_fail("null sequence");
    s->profile = on != 0;
    return 0;
}

int pcp_seq_profile_read(pcp_seq* s, float* ls_ms, int* launches, long long* points) {
    if (!s) return pcp_fail("null sequence");
    if (!ls_ms || !launches || !points) return pcp_fail("null output");
    if (s->in_flight[0] || s->in_flight[1]) return pcp_fail("collect the submissions in flight first");
    PCP_CHECK(hipSetDevice(s->device));
    PCP_CHECK(hipStreamSynchronize(s->compute));
    // launches after every pair has finished return at once: only those that did work count
    const int n = s->last_longest < s->prof_launches ? s->last_longest : s->prof_launches;
    float total = 0.f;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        PCP_CHECK(hipEventElapsedTime(&ms, s->prof_events[2 * i], s->prof_events[2 * i + 1]));
        total += ms;
    }
    *ls_ms = total;
    *launches = n;
    *points = s->last_points;
    return 0;
}

int pcp_seq_profile_read_launches(pcp_seq* s, float* ms_each, int capacity, int* launches) {
    if (!s) return pcp_fail("null sequence");
    if (!ms_each || !launches || capacity < 0) return pcp_fail("null output");
    if (s->in_flight[0] || s->in_flight[1]) return pcp_fail("collect the submissions in flight first");
    PCP_CHECK(hipSetDevice(s->device));
    PCP_CHECK(hipStreamSynchronize(s->compute));
    const int n = s->last_longest < s->prof_launches ? s->last_longest : s->prof_launches;
    for (int i = 0; i < n && i < capacity; ++i)
        PCP_CHECK(hipEventElapsedTime(&ms_each[i], s->prof_events[2 * i], s->prof_events[2 * i + 1]));
    *launches = n;
    return 0;
}

int pcp_host_alloc(size_t bytes, void** out) {
    if (!out) return pcp_fail("null out pointer");
    *out = nullptr;
    PCP_CHECK(hipHostMalloc(out, bytes ? bytes : 1));
    return 0;
}

int pcp_host_free(void* p) {
    if (p) PCP_CHECK(hipHostFree(p));
    return 0;
}

}  // extern "C"
