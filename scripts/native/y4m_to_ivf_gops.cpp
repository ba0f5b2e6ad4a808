// y4m_to_ivf_gops.cpp -- one YUV4MPEG2 file to one IVF file with its closed GOPs coded SIDE BY SIDE: the file-to-file form of what
// bench.py's headline measures, as a complete C++ user of the C ABI.
//   y4m_to_ivf_gops <in.y4m> <out.ivf> [-g gop] [-partitions P] [-qmin q] [-qmax q] [-SSIM-target t] [-altref-range n]
//                   [-no-check-ssim] [-conformant] [-chunks N (48)] [-batch B (6)]
// A key frame resets every reference (intra_part.h:1091-1098, inter_part.h:35-50), so the frames [k g, (k + 1) g) of a run with
// `-g g` are a unit of their own: N such chunks are in flight at once, B of them advance together as one batch (every stage ONE
// launch for the batch: vp8drv_batch_*), a host thread per batch; the input is streamed (two page-locked frame buffers per chunk in
// flight: the batch's thread reads frame t + 2 of its members while frame t + 1, already on its way to the device, is coded; a frame's
// planes lie end to end, one copy per frame: vp8hip_batch_upload_current / _prefetch_current), a writer thread writes every finished frame as soon as all
// frames before it are written (a frame's bytes are held only until then).  An error in one batch stops the others; a frame that
// check_SSIM sends back is reported (the file is then not the serial program's: see below).
// The file is, byte for byte, what `y4m_to_ivf -no-scene-detect -g g` (one video, frame after frame: the reference's loop) writes --
// as long as no frame is sent back by check_SSIM to be a key frame (the reference then restarts its GOP counter, vp8enc.cpp:443-453,
// intra_part.h:1091, and the serial run's later key frames move; with the default -SSIM-target -1 none is) and no scene detection
// is asked for (a cut moves the key frames the same way).  tests/test_gpu_file_roundtrip.py holds the two programs against each other.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <unistd.h>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "vp8hip_bitstream.h"
#include "vp8hip_driver.h"
#include "vp8hip_host.h"

#define CK(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, vp8hip_status_string(rc_)); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: see the head of y4m_to_ivf_gops.cpp\n"); return 2; }
    vp8drv_config cfg;
    vp8drv_default_config(&cfg);
    int in_flight = 48, batch = 6;
    for (int i = 3; i < argc; ++i) {
        auto val = [&]() { return i + 1 < argc ? argv[++i] : "0"; };
        if (!strcmp(argv[i], "-g")) cfg.gop_size = atoi(val());
        else if (!strcmp(argv[i], "-partitions")) cfg.num_partitions = atoi(val());
        else if (!strcmp(argv[i], "-qmin")) cfg.qi_min = atoi(val());
        else if (!strcmp(argv[i], "-qmax")) cfg.qi_max = atoi(val());
        else if (!strcmp(argv[i], "-SSIM-target")) cfg.ssim_target = (float)atof(val());
        else if (!strcmp(argv[i], "-altref-range")) cfg.altref_range = atoi(val());
        else if (!strcmp(argv[i], "-no-check-ssim")) cfg.check_ssim = 0;
        else if (!strcmp(argv[i], "-conformant")) cfg.conformant_stream = 1;
        else if (!strcmp(argv[i], "-chunks")) in_flight = atoi(val());
        else if (!strcmp(argv[i], "-batch")) batch = atoi(val());
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (cfg.gop_size < 1 || batch < 1 || batch > VP8HIP_MAX_BATCH || in_flight < 1) { fprintf(stderr, "bad -g / -batch / -chunks\n"); return 2; }
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    FILE *in = fopen(argv[1], "rb");
    if (!in) { perror(argv[1]); return 1; }
    uint8_t head[256];
    const size_t got = fread(head, 1, sizeof head, in);
    int32_t W = 0, H = 0, fps = 0;
    size_t first = 0;
    if (vp8host_y4m_parse_header(head, got, &W, &H, &fps, &first) != 0 || (W & 1) || (H & 1)) {
        fprintf(stderr, "%s: not a YUV4MPEG2 stream the reference accepts\n", argv[1]);
        return 1;
    }
    fseek(in, 0, SEEK_END);
    const size_t file_size = (size_t)ftell(in);
    const size_t ysz = (size_t)W * H, csz = ysz / 4, fsz = ysz + 2 * csz, rec = fsz + 6;      // a frame and the marker behind it (get_yuv420_frame, encIO.h:203-254)
    const int nframes = (int)((file_size - first + 6) / rec);
    if (nframes < 1) { fprintf(stderr, "%s: no frame\n", argv[1]); return 1; }
    fclose(in);
    // The input is STREAMED: every chunk in flight has two page-locked frame buffers, and the thread of its batch reads frame t + 2
    // of its members into the one while frame t + 1 (in the other, already on its way to the device) is coded -- the file is read by
    // as many threads as there are batches, beside the device's work, and no more of it is in memory than the chunks need
    const int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { perror(argv[1]); return 1; }
    const double t_read = now();
    const int Wc = (W + 15) / 16 * 16, Hc = (H + 15) / 16 * 16;      // video.wrk_*, init.h:375-392
    if (Wc != W || Hc != H) { cfg.src_width = W; cfg.src_height = H; }
    const int g = cfg.gop_size, nchunks = (nframes + g - 1) / g;
    if (in_flight > nchunks) in_flight = nchunks;
    const int nbatches = (in_flight + batch - 1) / batch;
    // a driver per chunk in flight, formed into batches as they are made (every batch then sits on a hardware queue of its own,
    // INTEGRATION.md section 6); driver j codes chunks j, j + in_flight, j + 2 in_flight ...: after g frames its own GOP counter is
    // at a key frame again
    std::vector<vp8drv *> drv((size_t)in_flight, nullptr);
    std::vector<vp8drv_batch *> bat((size_t)nbatches, nullptr);
    for (int j = 0; j < in_flight; ++j) {
        CK(vp8drv_create(&drv[j], Wc, Hc, 0, &cfg));
        CK(vp8hip_reserve_frame_path_dense(vp8drv_context(drv[j])));      // frame t + 1 is started before frame t's bytes are taken: no frame is ever coded twice
        if ((j + 1) % batch == 0 || j == in_flight - 1) {
            const int k = j / batch, j0 = k * batch;
            CK(vp8drv_batch_create(&bat[k], &drv[j0], j - j0 + 1));
        }
    }
    uint8_t *pinned = nullptr;
    CK(vp8hip_host_alloc(0, (size_t)in_flight * 2 * rec, reinterpret_cast<void **>(&pinned)));
    const double t_made = now();
    if (cfg.ssim_target > 0.0f)
        fprintf(stderr, "y4m_to_ivf_gops: -SSIM-target %.2f: a frame that check_SSIM sends back to be a key frame restarts the serial program's GOP counter "
                        "(vp8enc.cpp:443-453); if that happens this file is a valid stream but NOT byte for byte `y4m_to_ivf -g %d`'s (reported below)\n", cfg.ssim_target, g);
    FILE *out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 1; }
    uint8_t fh[32];
    // the reference's file says one frame more than it holds (encIO.h:124-134, vp8enc.cpp:487-489; REFERENCE_DEFECTS.md #8) -- reproduced
    fwrite(fh, 1, vp8bs_ivf_file_header(fh, W, H, (uint32_t)(fps ? fps : 30), 1, (uint32_t)nframes + 1), out);
    std::vector<std::vector<uint8_t>> out_frames((size_t)nframes);
    std::vector<char> ready((size_t)nframes, 0);
    std::mutex wm;
    std::condition_variable wcv;
    std::atomic<bool> failed{false};
    std::atomic<int> resent{0};
    size_t total = 32;
    std::thread writer([&] {      // frames leave in order as soon as every frame before them has left
        for (int t = 0; t < nframes; ++t) {
            std::vector<uint8_t> f;
            {
                std::unique_lock<std::mutex> l(wm);
                wcv.wait(l, [&] { return ready[(size_t)t] || failed.load(); });
                if (!ready[(size_t)t]) return;
                f.swap(out_frames[(size_t)t]);
            }
            uint8_t ph[12];
            fwrite(ph, 1, vp8bs_ivf_frame_header(ph, (uint32_t)f.size(), (uint32_t)t), out);
            fwrite(f.data(), 1, f.size(), out);
            total += 12 + f.size();
        }
    });
    std::vector<int> rc((size_t)nbatches, VP8HIP_OK), keys((size_t)nbatches, 0);
    const size_t cap = (size_t)(Wc / 16) * (Hc / 16) * 1900 + (1 << 20);
    std::vector<std::thread> th;
    for (int k = 0; k < nbatches; ++k)
        th.emplace_back([&, k] {
            if (k) std::this_thread::sleep_for(std::chrono::microseconds(200 * k));      // batches that start together stay in lockstep (vp8hip_driver.h)
            const int j0 = k * batch, n = (k == nbatches - 1 ? in_flight - j0 : batch);
            std::vector<uint8_t> buf(cap);
            const void *y[VP8HIP_MAX_BATCH], *u[VP8HIP_MAX_BATCH], *v[VP8HIP_MAX_BATCH];
            const uint8_t *py[VP8HIP_MAX_BATCH], *pu[VP8HIP_MAX_BATCH], *pv[VP8HIP_MAX_BATCH];
            int on[VP8HIP_MAX_BATCH], on_next[VP8HIP_MAX_BATCH], force[VP8HIP_MAX_BATCH], was_key[VP8HIP_MAX_BATCH];
            for (int round = 0; rc[k] == VP8HIP_OK && !failed.load() && (size_t)round * in_flight + j0 < (size_t)nchunks; ++round) {
                // member i codes chunk c_i = round * in_flight + j0 + i: frames c_i g + t, t < g
                auto frame_of = [&](int i, int t) { const long c = (long)round * in_flight + j0 + i; return c < nchunks && c * g + t < nframes ? (int)(c * g + t) : -1; };
                auto slot = [&](int i, int t) { return pinned + ((size_t)(j0 + i) * 2 + (size_t)(t & 1)) * rec; };
                auto load = [&](int t) {      // frame t of every member's chunk from the file into the member's buffer t & 1 (get_yuv420_frame, encIO.h:203-254)
                    for (int i = 0; i < n; ++i) {
                        const int f = frame_of(i, t);
                        if (f < 0) continue;
                        const size_t want = f + 1 < nframes ? rec : fsz;          // ... with the next frame's marker, if there is one
                        if (pread(fd, slot(i, t), want, (off_t)(first + (size_t)f * rec)) != (ssize_t)want) return VP8HIP_ERR_ARG;
                        if (want == rec && !vp8host_y4m_frame_marker_ok(slot(i, t) + fsz)) return VP8HIP_ERR_FORMAT;
                    }
                    return VP8HIP_OK;
                };
                auto planes = [&](int t, int *members) {
                    int any = 0;
                    for (int i = 0; i < n; ++i) {
                        const int f = frame_of(i, t);
                        members[i] = f >= 0;
                        any |= members[i];
                        const uint8_t *p = f >= 0 ? slot(i, t) : nullptr;
                        y[i] = py[i] = p; u[i] = pu[i] = p ? p + ysz : nullptr; v[i] = pv[i] = p ? p + ysz + csz : nullptr;
                    }
                    return any;
                };
                // the loop of vp8drv_batches_encode_frames_host with the bytes kept: frame t + 1 enqueued before frame t's bytes are waited for
                if ((rc[k] = load(0)) != VP8HIP_OK) break;
                planes(0, on);
                for (int i = 0; i < n; ++i) force[i] = 1;
                rc[k] = vp8drv_batch_encode_frame_host(bat[k], on, y, u, v, force, was_key);
                if (rc[k] == VP8HIP_OK && g > 1 && planes(1, on_next) && (rc[k] = load(1)) == VP8HIP_OK)
                    rc[k] = vp8drv_batch_prefetch_frame_host(bat[k], py, pu, pv);      // frame 1 read and on its way
                for (int t = 0; t < g && rc[k] == VP8HIP_OK && !failed.load(); ++t) {
                    int cur_on[VP8HIP_MAX_BATCH];
                    if (!planes(t, cur_on)) break;
                    rc[k] = vp8drv_batch_get_frame_begin(bat[k], cur_on);        // frame t's type is final here (its verdict is in)
                    if (rc[k] != VP8HIP_OK) break;
                    for (int i = 0; i < n; ++i)
                        if (cur_on[i]) {
                            const int key = vp8drv_resolve(drv[j0 + i]) == 1;
                            keys[k] += key;
                            if (key && t > 0) ++resent;      // a frame inside a chunk ended as a key frame: check_SSIM sent it back
                        }
                    if (t + 1 < g && planes(t + 1, on_next)) {
                        for (int i = 0; i < n; ++i) force[i] = 0;
                        rc[k] = vp8drv_batch_encode_frame_host(bat[k], on_next, y, u, v, force, was_key);
                        // ... and the one after it read (into the buffer frame t came from: its upload has returned) and started on its way
                        if (rc[k] == VP8HIP_OK && t + 2 < g && planes(t + 2, on) && (rc[k] = load(t + 2)) == VP8HIP_OK)
                            rc[k] = vp8drv_batch_prefetch_frame_host(bat[k], py, pu, pv);
                        if (rc[k] != VP8HIP_OK) break;
                    }
                    for (int i = 0; i < n && rc[k] == VP8HIP_OK; ++i) {
                        if (!cur_on[i]) continue;
                        size_t size = 0;
                        rc[k] = vp8drv_get_frame_end(drv[j0 + i], buf.data(), buf.size(), &size);
                        if (rc[k] == VP8HIP_OK) {
                            const size_t f = (size_t)frame_of(i, t);
                            std::lock_guard<std::mutex> l(wm);
                            out_frames[f].assign(buf.begin(), buf.begin() + (long)size);
                            ready[f] = 1;
                        }
                    }
                    wcv.notify_all();
                }
            }
            if (rc[k] != VP8HIP_OK) {      // one batch failed: the others stop at their next frame, the writer at the first frame that never comes
                failed = true;
                wcv.notify_all();
            }
        });
    for (auto &t : th) t.join();
    const double t_coded = now();
    { std::lock_guard<std::mutex> l(wm); }
    wcv.notify_all();
    writer.join();
    fclose(out);
    const double t_written = now();
    int nkeys = 0, first_error = VP8HIP_OK;
    // one way out, with or without an error: batches, drivers, the page-locked block and the file descriptor are given back
    for (int k = 0; k < nbatches; ++k) vp8drv_batch_destroy(bat[k]);
    for (auto d : drv) vp8drv_destroy(d);
    for (int k = 0; k < nbatches; ++k) { nkeys += keys[k]; if (rc[k] != VP8HIP_OK && first_error == VP8HIP_OK) first_error = rc[k]; }
    vp8hip_host_free(0, pinned);
    close(fd);
    if (first_error != VP8HIP_OK) {
        fprintf(stderr, "y4m_to_ivf_gops: a batch failed: %d (%s); %s is incomplete\n", first_error, vp8hip_status_string(first_error), argv[2]);
        return 1;
    }
    if (resent.load())
        fprintf(stderr, "y4m_to_ivf_gops: check_SSIM sent %d frame(s) back to be key frames: the serial program restarts its GOP counter there, so this file is a "
                        "valid stream but NOT byte for byte `y4m_to_ivf -g %d`'s\n", resent.load(), g);
    printf("%s: %d frames %dx%d (coded %dx%d) in %d closed GOPs of %d, %d in flight in %d batches, %d key frames, %zu bytes; %d hardware queues\n", argv[2], nframes, W, H, Wc,
           Hc, nchunks, g, in_flight, nbatches, nkeys, total, vp8hip_hw_queues());
    printf("  seconds: opening %.3f, %d contexts, their scratch and frame buffers %.3f, reading + coding %.3f (%.0f frames/s, %.2f M macroblocks/s, every frame from the file and over the "
           "host-device link both ways), writing %.3f, tearing down %.3f\n", t_read - t_start, in_flight, t_made - t_read, t_coded - t_made, nframes / (t_coded - t_made),
           (double)nframes * (Wc / 16) * (Hc / 16) / (t_coded - t_made) / 1e6, t_written - t_coded, now() - t_written);
    return 0;
}
