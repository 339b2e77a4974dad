// sharded_host_example.cpp -- what a C++ multi-GPU host (one process or thread per GPU, td.cpp-style tracker loop) links and calls:
// the sharded device-resident frame as ONE native call with the caller's RCCL communicator.  Compiled by __graft_entry__.build()
// (hipcc, links libmot_amd only -- librccl is bound by the library at run time); run it under mpirun / one process per GPU with a
// broadcast ncclUniqueId.  It is an integration example, not a test: the tests are tests/test_gpu_nccl.py.
//
//   hipcc -std=c++17 -I include tools/sharded_host_example.cpp -L multiple-object-tracking_amd -lmot_amd -o sharded_host_example
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "mot_abi.h"

// the two RCCL entry points a host needs besides ours (declared here so the example builds without the rccl headers)
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
int ncclGetUniqueId(ncclUniqueId*);
int ncclCommInitRank(ncclComm_t*, int nranks, ncclUniqueId id, int rank);
}

// one rank's frame loop: frames and detection lists are already in this GPU's memory (capture / detector threads own that part)
int run_rank(int rank, int world, ncclComm_t comm, const void* const* frames_dev, const void* const* dets_dev, const int* n_dets, int n_frames)
{
    mot_config cfg; mot_config_default(&cfg);
    cfg.device = rank; cfg.tracker_kind = MOT_TRACKER_KCF; cfg.max_tracks = 1024; cfg.max_dets = 1024;
    cfg.rank = rank; cfg.world = world;                                 // tracks sharded round-robin / least-loaded rank
    mot_ctx* ctx = nullptr;
    if (mot_ctx_create(&cfg, &ctx) != MOT_OK) { std::fprintf(stderr, "rank %d: %s\n", rank, mot_last_error()); return 1; }
    for (int f = 0; f < n_frames; f++)                                  // predict -> ncclAllGather(bbox_t) -> association -> update, enqueued only
        if (mot_step_frame_sharded(ctx, frames_dev[f], dets_dev[f], n_dets[f], comm) != MOT_OK) { std::fprintf(stderr, "rank %d: %s\n", rank, mot_last_error()); return 1; }
    int n_live = 0;
    if (mot_live_count(ctx, &n_live) != MOT_OK) return 1;               // synchronises
    std::printf("rank %d: %d live tracks after %d frames\n", rank, n_live, n_frames);
    return mot_ctx_destroy(ctx);
}

int main() { std::puts("integration example: call run_rank() from one process per GPU (see INTEGRATION.md, section C)"); return 0; }
