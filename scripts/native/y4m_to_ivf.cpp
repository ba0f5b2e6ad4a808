// y4m_to_ivf.cpp -- the reference's program with the path swapped in, as a complete C++ user of the C ABI: YUV4MPEG2 in,
// IVF out (main() of src/vp8enc.cpp reduced to: parse the header, per frame read / code / write, patch the frame count).
//   y4m_to_ivf <in.y4m> <out.ivf> [-g gop] [-partitions P] [-qmin q] [-qmax q] [-SSIM-target t] [-altref-range n] [-no-scene-detect]
//              [-no-check-ssim] [-conformant]
// As in the reference, check_SSIM runs after every inter frame and scene_change() looks at every frame that would be an inter frame.
// Everything between the two files runs behind include/vp8hip_driver.h; the frames are handed over at their source size
// and padded on the device (cfg.src_width / src_height), key frames carry that size as the display size.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "vp8hip_bitstream.h"
#include "vp8hip_driver.h"
#include "vp8hip_host.h"

#define CK(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, vp8hip_status_string(rc_)); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: see the head of y4m_to_ivf.cpp\n"); return 2; }
    vp8drv_config cfg;
    vp8drv_default_config(&cfg);
    cfg.scene_detect = 1;                       // main() calls scene_change() for every would-be inter frame (vp8enc.cpp:408)
    cfg.overlap_filter = 1;
    for (int i = 3; i < argc; ++i) {
        auto val = [&]() { return i + 1 < argc ? argv[++i] : "0"; };
        if (!strcmp(argv[i], "-g")) cfg.gop_size = atoi(val());
        else if (!strcmp(argv[i], "-partitions")) cfg.num_partitions = atoi(val());
        else if (!strcmp(argv[i], "-qmin")) cfg.qi_min = atoi(val());
        else if (!strcmp(argv[i], "-qmax")) cfg.qi_max = atoi(val());
        else if (!strcmp(argv[i], "-SSIM-target")) cfg.ssim_target = (float)atof(val());
        else if (!strcmp(argv[i], "-altref-range")) cfg.altref_range = atoi(val());
        else if (!strcmp(argv[i], "-scene-detect")) cfg.scene_detect = 1;
        else if (!strcmp(argv[i], "-no-scene-detect")) cfg.scene_detect = 0;
        else if (!strcmp(argv[i], "-no-check-ssim")) cfg.check_ssim = 0;
        else if (!strcmp(argv[i], "-conformant")) cfg.conformant_stream = 1;
        else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
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
    fseek(in, (long)first, SEEK_SET);
    const int Wc = (W + 15) / 16 * 16, Hc = (H + 15) / 16 * 16;      // video.wrk_*, init.h:375-392
    if (Wc != W || Hc != H) { cfg.src_width = W; cfg.src_height = H; }
    vp8drv *drv = nullptr;
    CK(vp8drv_create(&drv, Wc, Hc, 0, &cfg));
    FILE *out = fopen(argv[2], "wb");
    if (!out) { perror(argv[2]); return 1; }
    uint8_t fh[32];
    fwrite(fh, 1, vp8bs_ivf_file_header(fh, W, H, (uint32_t)(fps ? fps : 30), 1, 0), out);     // frame count patched at the end (encIO.h:100-139)
    const size_t ysz = (size_t)W * H, csz = ysz / 4;
    std::vector<uint8_t> bytes((size_t)(Wc / 16) * (Hc / 16) * 1900 + (1 << 20));
    // A reader thread keeps a ring of page-locked frame buffers filled ahead of the coder (get_yuv420_frame's fread, encIO.h:204-254, off the
    // frame loop's thread: 3 MB per 1080p frame is 0.4 ms of the loop's 0.5), the frame after the one under way is started on its way to the
    // device (vp8drv_prefetch_frame_host) and, once the frame just coded has its verdict and its entropy stage enqueued, handed over
    // altogether (vp8drv_stage_frame_host: the pack, scene_change()'s scan) while that frame's loop filter still has most of its time to run.
    const size_t fsz = ysz + 2 * csz;
    enum { RING = 6 };
    uint8_t *buf[RING];
    int state[RING];                 // get_yuv420_frame's verdict on the frame in this slot: 1 = a frame, 0 = end of stream, -1 = broken
    for (int k = 0; k < RING; ++k) CK(vp8hip_host_alloc(0, fsz, reinterpret_cast<void **>(&buf[k])));
    std::mutex m;
    std::condition_variable cv;
    size_t rd_head = 0, rd_tail = 0, rd_freed = 0;      // filled by the reader / taken by the loop / given back by the loop
    bool stop = false;
    std::thread reader([&] {
        for (;;) {
            size_t slot;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return stop || rd_head - rd_freed < RING; });
                if (stop) return;
                slot = rd_head % RING;
            }
            int st = 1;
            if (fread(buf[slot], 1, fsz, in) != fsz) st = 0;
            else {
                uint8_t marker[6];
                const size_t n = fread(marker, 1, 6, in);
                if (n > 0 && !vp8host_y4m_frame_marker_ok(marker)) st = -1;
            }
            {
                std::lock_guard<std::mutex> l(m);
                state[slot] = st;
                ++rd_head;
            }
            cv.notify_all();
            if (st != 1) return;
        }
    });
    auto stop_reader = [&] {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv.notify_all();
        if (reader.joinable()) reader.join();
    };
    auto peek = [&](bool wait) -> int {          // the state of the next frame in the ring: 1 / 0 / -1, or 2 = not read yet (wait == false)
        std::unique_lock<std::mutex> l(m);
        if (wait) cv.wait(l, [&] { return rd_tail < rd_head; });
        return rd_tail < rd_head ? state[rd_tail % RING] : 2;
    };
    auto planes = [&](size_t seq, const uint8_t *p[3]) { p[0] = buf[seq % RING]; p[1] = p[0] + ysz; p[2] = p[1] + csz; };
    uint32_t n = 0, keys = 0;
    size_t total = 32;
    // One video: the loop filter of a frame runs beside the next frame's input side (vp8hip_filter_overlap), and its entropy stage
    // beside both on a third stream -- so frame t + 1 is read and started BEFORE frame t's bytes are taken (vp8drv_get_frame_begin /
    // _end).  The scratch is sized for the densest frame there can be: no frame is ever coded twice.
    CK(vp8hip_reserve_frame_path_dense(vp8drv_context(drv)));
    bool pending = false;
    size_t prefetched = (size_t)-1;
    auto prefetch_next = [&]() -> int {           // the frame at the ring's rd_tail started on its way, once
        if (peek(false) != 1 || prefetched == rd_tail) return 0;
        const uint8_t *p[3];
        planes(rd_tail, p);
        prefetched = rd_tail;
        return vp8drv_prefetch_frame_host(drv, p[0], p[1], p[2]);
    };
    const auto t_loop = std::chrono::steady_clock::now();      // (the frame loop by the host's clock: what scripts/drop_in_bench.py quotes as fps_loop)
    for (;;) {
        const int have = peek(true);
        if (have < 0) { fprintf(stderr, "broken stream!\n"); stop_reader(); return 1; }
        const bool got = have > 0;
        if (got) {
            const uint8_t *p[3];
            planes(rd_tail, p);
            {   // the previous frame's buffer goes back to the reader; this one is taken
                std::lock_guard<std::mutex> l(m);
                if (rd_tail > rd_freed) rd_freed = rd_tail;
                ++rd_tail;
            }
            cv.notify_all();
            CK(vp8drv_encode_frame_host(drv, p[0], p[1], p[2], 0));      // (uploads nothing if the frame was handed over early)
            CK(prefetch_next());
        }
        if (pending) {      // the previous frame's bytes
            size_t size = 0;
            CK(vp8drv_get_frame_end(drv, bytes.data(), bytes.size(), &size));
            uint8_t ph[12];
            fwrite(ph, 1, vp8bs_ivf_frame_header(ph, (uint32_t)size, n), out);
            fwrite(bytes.data(), 1, size, out);
            total += 12 + size;
            ++n;
            pending = false;
        }
        if (!got) break;
        CK(vp8drv_get_frame_begin(drv));
        const int key = vp8drv_resolve(drv);       // the frame's final type: check_SSIM may have sent it back to be a key frame
        CK(key);
        keys += key;
        pending = true;
        if (peek(false) == 1) {     // the next frame, early: current on the device, its scene scan under way, the one after it on its way
            const uint8_t *p[3];
            planes(rd_tail, p);
            CK(vp8drv_stage_frame_host(drv, p[0], p[1], p[2]));
        }
    }
    stop_reader();
    const double loop_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count();
    fseek(out, 0, SEEK_SET);
    // the reference's file says one frame more than it holds: write_output_header counts from a frame number that main() has
    // already advanced past the last frame (encIO.h:124-134, vp8enc.cpp:487-489; REFERENCE_DEFECTS.md #8) -- reproduced, the bar
    // being the reference's bytes
    fwrite(fh, 1, vp8bs_ivf_file_header(fh, W, H, (uint32_t)(fps ? fps : 30), 1, n + 1), out);
    fclose(out);
    fclose(in);
    vp8drv_stats st;
    vp8drv_get_stats(drv, &st);
    vp8drv_destroy(drv);
    for (int k = 0; k < RING; ++k) vp8hip_host_free(0, buf[k]);
    printf("%s: %u frames %dx%d (coded %dx%d), %u key (%d by scene change, %d recoded), %zu bytes; %d hardware queues\n", argv[2], n, W, H, Wc, Hc, keys,
           st.scene_changes, st.redone_as_key, total, vp8hip_hw_queues());
    printf("%.6f s of reading + coding + writing (%.1f frames/s)\n", loop_s, n / (loop_s > 0 ? loop_s : 1));
    return 0;
}
