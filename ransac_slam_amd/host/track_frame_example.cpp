// track_frame_example.cpp -- the five hot calls of System::TrackRunning (System.cpp:117-129)
// driven through the host-side mirror (ransac_slam_hip.hpp).  Reads a frame dumped by
// tests/test_host_adapter.py, writes the outputs; the test compares them with the oracle.
//   usage: track_frame_example frame.bin out.bin [map.bin]
// With a third argument the map management of the next frame follows on the device (Map mirror):
// delete, convert, insert, then ExtendKF::ekf_prediction with nothing uploaded.
#include <cstdio>
#include <cstring>

#include "ransac_slam_hip.hpp"

using namespace ransac_slam_hip;

template <typename T>
static bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s frame.bin out.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("frame"); return 2; }
    int32_t hdr[4];   // n, L, n_draws, compat
    if (!rd(f, hdr, 4)) return 2;
    const int n = hdr[0], L = hdr[1], nd = hdr[2], compat = hdr[3];
    std::vector<uint8_t> type(L), ic(L);
    std::vector<double> x(n), P((size_t)n * n), z(2 * (size_t)L), draws(nd);
    if (!rd(f, type.data(), L) || !rd(f, ic.data(), L) || !rd(f, x.data(), n) || !rd(f, P.data(), (size_t)n * n) ||
        !rd(f, z.data(), 2 * (size_t)L) || !rd(f, draws.data(), nd)) return 2;
    fclose(f);

    CamParam cam{0.06333, 0.01390, 240, 320, 1.7945 / 0.0112, 1.4433 / 0.0112, 2.1735, 0.0112, 0.0112};
    try {
        ExtendKF kf(&cam, 0, compat, nd);
        Tracking tracking(&kf);
        kf.features_info.resize(L);
        for (int i = 0; i < L; ++i) kf.features_info[i].type = type[i] ? "cartesian" : "inversedepth";
        kf.x_k_km1.v = x;                    // what ekf_prediction() leaves behind (System.cpp:114)
        kf.p_k_km1.resize(n, n); kf.p_k_km1.v = P;

        tracking.search_IC_matches_predict();                       // System.cpp:117 (first half)
        for (int i = 0; i < L; ++i) {                               // stand-in for matching(), Tracking.cpp:343-347
            Feature& ft = kf.features_info[i];
            if (ic[i] && ft.h.rows()) { ft.individually_compatible = true; ft.z.resize(2); ft.z(0) = z[2 * i]; ft.z(1) = z[2 * i + 1]; }
        }
        tracking.replay_draws = draws;                              // the test's draw list instead of std::rand
        tracking.ransac_hypotheses();                               // System.cpp:120
        kf.ekf_update_li_inliers();                                  // System.cpp:123
        tracking.rescue_hi_inliers();                                // System.cpp:126
        kf.ekf_update_hi_inliers(true);                              // System.cpp:129

        FILE* o = fopen(argv[2], "wb");
        if (!o) { perror("out"); return 2; }
        int32_t sc[3] = {tracking.best_hyp, tracking.best_support, tracking.hyps_evaluated};
        fwrite(sc, sizeof(int32_t), 3, o);
        std::vector<uint8_t> li(L), hi(L), has_h(L);
        std::vector<double> h(2 * (size_t)L, 0.0), S(4 * (size_t)L, 0.0);
        for (int i = 0; i < L; ++i) {
            const Feature& ft = kf.features_info[i];
            li[i] = ft.low_innovation_inlier; hi[i] = ft.high_innovation_inlier; has_h[i] = ft.h.rows() ? 1 : 0;
            if (has_h[i]) { h[2 * i] = ft.h(0); h[2 * i + 1] = ft.h(1); for (int q = 0; q < 4; ++q) S[4 * i + q] = ft.S.v[q]; }
        }
        fwrite(li.data(), 1, L, o); fwrite(hi.data(), 1, L, o); fwrite(has_h.data(), 1, L, o);
        fwrite(h.data(), sizeof(double), h.size(), o); fwrite(S.data(), sizeof(double), S.size(), o);
        fwrite(kf.x_k_k.data(), sizeof(double), n, o);
        fwrite(kf.p_k_k.data(), sizeof(double), (size_t)n * n, o);
        fclose(o);

        if (argc > 3) {                                              // System.cpp:111-114 of the next frame
            Map map(&kf);
            kf.features_info.erase(kf.features_info.begin() + 2);   // Map.cpp:27-28
            map.delete_a_feature(3);
            const int converted = map.inversedepth_2_cartesian(1e-30);   // nothing is that linear: no edit
            const size_t visible = map.predicted_positions().size();
            const double uv[2] = {140.0, 100.0};
            map.add_a_feature(uv);                                   // no image-side record here: state surgery only
            std::vector<uint8_t> t2; std::vector<int32_t> o2;
            int off = 13;
            for (const Feature& ft : kf.features_info) {
                const bool id = ft.type == "inversedepth";
                t2.push_back(id ? 0 : 1); o2.push_back(off); off += id ? 6 : 3;
            }
            kf.ekf_prediction(false, t2, o2);
            std::vector<double> xp(off), Pp((size_t)off * off);
            const int rc = rslam_fetch_prior(kf.ctx(), xp.data(), Pp.data());
            if (rc) throw Error(rc, "rslam_fetch_prior");
            FILE* m = fopen(argv[3], "wb");
            if (!m) { perror("map"); return 2; }
            int32_t mh[4] = {off, (int32_t)t2.size(), converted, (int32_t)visible};
            fwrite(mh, sizeof(int32_t), 4, m);
            fwrite(t2.data(), 1, t2.size(), m);
            fwrite(xp.data(), sizeof(double), xp.size(), m);
            fwrite(Pp.data(), sizeof(double), Pp.size(), m);
            fclose(m);
        }
    } catch (const Error& e) {
        fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
