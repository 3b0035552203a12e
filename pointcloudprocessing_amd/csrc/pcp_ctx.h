// Context object behind the opaque pcp_ctx handle, shared by the translation
// units of libpcp.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#define PCP_BLOCK 64   // threads per block of the search kernels
#ifndef PCP_LS_THREADS
#define PCP_LS_THREADS 256         // threads per block of the hinted launches of a batch (k_ls_batch): the block shares its ring searches
#endif

struct IcpState {
    double R[9];
    double t[3];
    int pending;    // (R, t) still has to be applied to the source points
    int iter;       // dispatches executed so far
    int done;
    int converged;
    int singular;
    int pad[3];
};

struct pcp_voxel_state;   // device side of the voxel map, pcp_voxel.hip
void pcp_voxel_state_free(pcp_voxel_state* v);

struct pcp_ctx {
    int device = 0, max_points = 0, max_scans = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    float4 *a = nullptr, *ua = nullptr, *normals = nullptr, *b32 = nullptr;
    double *b_in = nullptr, *b64 = nullptr;
    float* hint = nullptr;
    int hinted_upto = 0;               // source points [0, hinted_upto) have been searched since the hints were last reset
    uint32_t* nn_idx = nullptr;        // 2 max_points words: the indices of the last search, its squared distances behind them
    int2* scan = nullptr;
    int2* scan_host = nullptr;     // page-locked staging of the scan-line table (max_scans rows)
    hipEvent_t scan_copied = nullptr;   // recorded behind the copy out of scan_host
    // page-locked staging of the stepwise calls (made on first use): the source cloud on its way up, and the
    // correspondence buffer (index, squared distance, ring hint per point) on its way down
    float4* stage_in = nullptr;
    uint32_t* stage_idx = nullptr;     // indices, squared distances behind them
    float* stage_hint = nullptr;
    struct pcp_batch* batch = nullptr;  // the batch that has adopted this context, if any
    float4* ring_stat = nullptr;   // per ring: (sin, cos) of lowest and highest elevation
    float4* ring_bins = nullptr;   // the same per ring and azimuth bin (PCP_AZ_BINS per ring), then the range bands of the same points
    float *partials = nullptr, *lvl0 = nullptr, *lvl1 = nullptr;
    float *hb_trace = nullptr, *dx_trace = nullptr;   // max_iter_cap x 64 / x 6
    IcpState* state = nullptr;
    void* pose_steps = nullptr;      // scratch of pcp_pose_from_steps (lazily allocated)
    double* pose = nullptr;          // 16 doubles: t_opt[3], R_opt[9], rank flag, steps applied (pose kernel)
    float *tiles = nullptr, *tiles2 = nullptr;        // lazily allocated
    float4 origin = {0, 0, 0, 0};
    int n_a = 0, n_b = 0, n_scans = 0;
    int n_scans_search = 0;        // rings [n_scans_search, n_scans) repeat ring n_scans_search - 1 (padding)
    bool have_target = false, have_normals = false;
    bool have_pose = false;        // a loop has run and left its pose in `pose`
    int run_mode = 0, run_max_iter = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t nn_ev0 = nullptr, nn_ev1 = nullptr;   // around the kernels of the last pcp_nn (created on first use)
    bool nn_timed = false;
    bool profile = false;                 // bracket every LS launch of the loop with events
    std::vector<hipEvent_t> prof_events;  // pairs (before, after), created on demand
    int prof_used = 0;
    pcp_voxel_state* vox = nullptr;      // created by the first voxel call
    static const int max_iter_cap = 1024;
};

