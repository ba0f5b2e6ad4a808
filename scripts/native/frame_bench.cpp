// frame_bench.cpp -- the native frame loop driven from C++ (no interpreter between the calls): S GOP chunks, each its own
// vp8drv, either one host thread per chunk (blocking vp8drv_get_frame) or ONE host thread for all of them
// (vp8drv_get_frame_begin / _end).  Frames come from a raw I420 file (scripts/native/run_frame_bench.sh writes one
// from the synthetic sequence the Python benches use).
//   frame_bench <i420 file> <W> <H> <streams> <frames> <partitions> <mode: threads|pipeline> <bitstream: 0|1> [check_ssim] [overlap_filter]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "vp8hip_driver.h"

#define CK(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, vp8hip_status_string(rc_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    if (argc < 9) { fprintf(stderr, "usage: see the head of frame_bench.cpp\n"); return 2; }
    const char *path = argv[1];
    const int W = atoi(argv[2]), H = atoi(argv[3]), S = atoi(argv[4]), N = atoi(argv[5]), P = atoi(argv[6]);
    const bool pipeline = !strcmp(argv[7], "pipeline"), emit = atoi(argv[8]) != 0;
    const int check = argc > 9 ? atoi(argv[9]) : 0, overlap = argc > 10 ? atoi(argv[10]) : 0;
    const size_t ysz = (size_t)W * H, csz = ysz / 4, fsz = ysz + 2 * csz;
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    fseek(f, 0, SEEK_END);
    const int nd = (int)(ftell(f) / (long)fsz);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> host(fsz * nd);
    if (fread(host.data(), fsz, nd, f) != (size_t)nd) return 1;
    fclose(f);
    uint8_t *dev = nullptr;
    if (hipMalloc(&dev, fsz * nd) != hipSuccess || hipMemcpy(dev, host.data(), fsz * nd, hipMemcpyHostToDevice) != hipSuccess) return 1;
    vp8drv_config cfg;
    vp8drv_default_config(&cfg);
    cfg.gop_size = 1 << 30;
    cfg.num_partitions = P;
    cfg.check_ssim = check;
    cfg.overlap_filter = overlap;
    std::vector<vp8drv *> d(S);
    for (int k = 0; k < S; ++k) CK(vp8drv_create(&d[k], W, H, 0, &cfg));
    const size_t cap = (size_t)(W / 16) * (H / 16) * 900 + 65536;
    std::vector<std::vector<uint8_t>> out(S, std::vector<uint8_t>(cap));
    std::vector<size_t> bytes(S, 0);
    auto frame_of = [&](int k, int t, const uint8_t *&y, const uint8_t *&u, const uint8_t *&v) {
        y = dev + fsz * (size_t)((t + 3 * k) % nd);
        u = y + ysz;
        v = u + csz;
    };
    auto run = [&](int n) {
        for (auto &b : bytes) b = 0;
        const auto t0 = std::chrono::steady_clock::now();
        if (pipeline) {
            for (int t = 0; t < n; ++t) {
                for (int k = 0; k < S; ++k) {
                    const uint8_t *y, *u, *v;
                    frame_of(k, t, y, u, v);
                    CK(vp8drv_encode_frame_device(d[k], y, u, v, 0));
                    if (emit) CK(vp8drv_get_frame_begin(d[k]));
                }
                if (emit)
                    for (int k = 0; k < S; ++k) {
                        size_t sz = 0;
                        CK(vp8drv_get_frame_end(d[k], out[k].data(), cap, &sz));
                        bytes[k] += sz;
                    }
            }
        } else {
            std::vector<std::thread> th;
            for (int k = 0; k < S; ++k)
                th.emplace_back([&, k] {
                    for (int t = 0; t < n; ++t) {
                        const uint8_t *y, *u, *v;
                        frame_of(k, t, y, u, v);
                        CK(vp8drv_encode_frame_device(d[k], y, u, v, 0));
                        if (emit) {
                            size_t sz = 0;
                            CK(vp8drv_get_frame(d[k], out[k].data(), cap, &sz));
                            bytes[k] += sz;
                        }
                    }
                    CK(vp8hip_synchronize(vp8drv_context(d[k])));
                });
            for (auto &t : th) t.join();
        }
        for (int k = 0; k < S; ++k) CK(vp8hip_synchronize(vp8drv_context(d[k])));
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    run(4);
    const double el = run(N);
    size_t total = 0;
    for (auto b : bytes) total += b;
    const double fps = (double)S * N / el, mbs = (double)(W / 16) * (H / 16);
    printf("%dx%d %2d streams x %d frames, %s, bitstream %s: %8.1f fps, %6.2f M MB/s, %.3f ms per frame device-wide", W, H, S, N,
           pipeline ? "one host thread " : "thread per chunk", emit ? "on " : "off", fps, fps * mbs / 1e6, el / ((double)S * N) * 1e3);
    if (emit) printf(", %.1f KiB per frame", (double)total / ((double)S * N) / 1024.0);
    printf("\n");
    for (auto p : d) vp8drv_destroy(p);
    (void)hipFree(dev);
    return 0;
}
