// shard_frame_example.cpp -- one rank of the hypothesis-sharded frame from C++: what the body of
// System::TrackRunning (System.cpp:117-129) becomes when the node runs one process per GPU.  Every rank loads the
// same frame, scores its slice of the draw list, the supports cross the xGMI links in ONE ncclAllGather inside
// rslam_shard_frame, and every rank ends with the same posterior.
//   usage: shard_frame_example frame.bin out.bin [rank world idfile [device [allreduce]]]
// A seventh argument "allreduce" takes north_star's literal collective instead: the slice's best hypothesis as ONE 8-byte key,
// ncclAllReduce(MAX) inside rslam_shard_frame_allreduce (no adaptive stop in that form: every draw is evaluated).
// rank 0 writes the ncclUniqueId to `idfile`, the other ranks wait for it (any shared file system will do; an MPI or
// ROS launch would broadcast it instead).  With no rank arguments: world = 1, the all-gather runs on a communicator
// of one rank (what the single-GPU test checks).  Frame / output files as track_frame_example.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

#include <rccl/rccl.h>

#include "ransac_slam_hip.hpp"

using namespace ransac_slam_hip;

template <typename T>
static bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }

#define CHK(call) do { const int rc_ = (call); if (rc_) { fprintf(stderr, "%s: %s\n", #call, rslam_error_string(rc_)); return 1; } } while (0)
#define NCHK(call) do { const ncclResult_t r_ = (call); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(r_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s frame.bin out.bin [rank world idfile [device]]\n", argv[0]); return 2; }
    const int rank = argc > 3 ? atoi(argv[3]) : 0, world = argc > 4 ? atoi(argv[4]) : 1;
    const char* idfile = argc > 5 ? argv[5] : nullptr;
    const int device = argc > 6 ? atoi(argv[6]) : rank;
    const bool by_allreduce = argc > 7 && strcmp(argv[7], "allreduce") == 0;
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !idfile)) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("frame"); return 2; }
    int32_t hdr[4];   // n, L, n_draws, compat
    if (!rd(f, hdr, 4)) return 2;
    const int n = hdr[0], L = hdr[1], nd = hdr[2], compat = hdr[3];
    std::vector<uint8_t> type(L), ic(L), vis(L);
    std::vector<int32_t> offset(L);
    std::vector<double> x(n), P((size_t)n * n), z(2 * (size_t)L), draws(nd);
    if (!rd(f, type.data(), L) || !rd(f, ic.data(), L) || !rd(f, x.data(), n) || !rd(f, P.data(), (size_t)n * n) ||
        !rd(f, z.data(), 2 * (size_t)L) || !rd(f, draws.data(), nd)) return 2;
    fclose(f);
    int off = 13;
    for (int i = 0; i < L; ++i) { offset[i] = off; off += type[i] ? 3 : 6; }

    CamParam cam{0.06333, 0.01390, 240, 320, 1.7945 / 0.0112, 1.4433 / 0.0112, 2.1735, 0.0112, 0.0112};
    try {
        ExtendKF kf(&cam, device, compat, nd, false, !by_allreduce);       // creates the context on this rank's GPU (and selects the device)
        rslam_ctx* ctx = kf.ctx();

        // the communicator of the node: one rank per process / GPU
        ncclUniqueId id;
        if (rank == 0) {
            NCHK(ncclGetUniqueId(&id));
            if (idfile) {
                std::string tmp = std::string(idfile) + ".tmp";
                FILE* o = fopen(tmp.c_str(), "wb");
                if (!o || fwrite(&id, sizeof(id), 1, o) != 1) return 2;
                fclose(o);
                rename(tmp.c_str(), idfile);
            }
        } else {
            FILE* in = nullptr;
            for (int tries = 0; tries < 600 && !(in = fopen(idfile, "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            if (!in || fread(&id, sizeof(id), 1, in) != 1) { fprintf(stderr, "no unique id in %s\n", idfile); return 2; }
            fclose(in);
        }
        ncclComm_t comm;
        NCHK(ncclCommInitRank(&comm, world, id, rank));

        // the frame, replicated on every rank; IC flags gated by the device's own visibility test (as matching() would)
        rslam_layout lay{n, L, type.data(), offset.data()};
        CHK(rslam_load_frame(ctx, &lay, x.data(), P.data(), z.data(), ic.data(), draws.data(), nd));
        CHK(rslam_step_predict(ctx));
        CHK(rslam_sync(ctx));
        std::vector<double> h(2 * (size_t)L, 0.0), S(4 * (size_t)L, 0.0);
        CHK(rslam_fetch_prediction(ctx, h.data(), vis.data(), S.data()));
        for (int i = 0; i < L; ++i) ic[i] = ic[i] && vis[i];
        CHK(rslam_load_measurements(ctx, z.data(), ic.data(), draws.data(), nd));

        // System.cpp:120-129 on this rank: its slice of the hypotheses, the all-gather, consensus + both updates
        for (int rep = 0; rep < 3; ++rep)           // replays start from the same resident prior (hipGraph from the 2nd on)
            CHK(by_allreduce ? rslam_shard_frame_allreduce(ctx, comm, rank, world, 1) : rslam_shard_frame(ctx, comm, rank, world, 1));
        CHK(rslam_sync(ctx));

        std::vector<double> x_new(n), P_new((size_t)n * n);
        std::vector<uint8_t> li(L), hi(L);
        int32_t sc[3], n_li = 0, n_hi = 0;
        CHK(rslam_fetch_results(ctx, x_new.data(), li.data(), hi.data(), &sc[0], &sc[1], &sc[2], &n_li, &n_hi));
        CHK(rslam_fetch_cov(ctx, P_new.data()));
        NCHK(ncclCommDestroy(comm));

        FILE* o = fopen(argv[2], "wb");
        if (!o) { perror("out"); return 2; }
        fwrite(sc, sizeof(int32_t), 3, o);
        fwrite(li.data(), 1, L, o); fwrite(hi.data(), 1, L, o); fwrite(vis.data(), 1, L, o);
        fwrite(h.data(), sizeof(double), 2 * (size_t)L, o); fwrite(S.data(), sizeof(double), 4 * (size_t)L, o);
        fwrite(x_new.data(), sizeof(double), n, o); fwrite(P_new.data(), sizeof(double), (size_t)n * n, o);
        fclose(o);
        printf("rank %d/%d: best_hyp %d support %d evaluated %d, %d low- and %d high-innovation inliers\n", rank, world,
               sc[0], sc[1], sc[2], n_li, n_hi);
    } catch (const Error& e) {
        fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
