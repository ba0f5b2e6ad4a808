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
    // two page-locked frame buffers: while frame t is coded, frame t + 1 is read and started on its way to the device
    // (vp8drv_prefetch_frame_host), so the copy never stands in front of a frame's first launch
    const size_t fsz = ysz + 2 * csz;
    uint8_t *buf[2] = {nullptr, nullptr};
    for (int k = 0; k < 2; ++k) CK(vp8hip_host_alloc(0, fsz, reinterpret_cast<void **>(&buf[k])));
    auto read_frame = [&](uint8_t *dst) -> int {       // get_yuv420_frame, encIO.h:203-254: 1 = a frame, 0 = end of stream, -1 = broken
        if (fread(dst, 1, fsz, in) != fsz) return 0;
        uint8_t marker[6];
        const size_t m = fread(marker, 1, 6, in);
        return (m > 0 && !vp8host_y4m_frame_marker_ok(marker)) ? -1 : 1;
    };
    uint32_t n = 0, keys = 0;
    size_t total = 32;
    // One video: the loop filter of a frame runs beside the next frame's input side (vp8hip_filter_overlap), and its entropy stage
    // beside both on a third stream -- so frame t + 1 is read and started BEFORE frame t's bytes are taken (vp8drv_get_frame_begin /
    // _end).  The scratch is sized for the densest frame there can be: no frame is ever coded twice.
    CK(vp8hip_reserve_frame_path_dense(vp8drv_context(drv)));
    bool pending = false;
    int cur = 0;
    const auto t_loop = std::chrono::steady_clock::now();      // (the frame loop by the host's clock: what scripts/drop_in_bench.py quotes as fps_loop)
    int have = read_frame(buf[cur]);
    if (have < 0) { fprintf(stderr, "broken stream!\n"); return 1; }
    for (;;) {
        const bool got = have > 0;
        if (got) {
            uint8_t *f = buf[cur];
            CK(vp8drv_encode_frame_host(drv, f, f + ysz, f + ysz + csz, 0));
            cur ^= 1;
            have = read_frame(buf[cur]);               // the next frame, read while this one is coded ...
            if (have < 0) { fprintf(stderr, "broken stream!\n"); return 1; }
            if (have > 0) CK(vp8drv_prefetch_frame_host(drv, buf[cur], buf[cur] + ysz, buf[cur] + ysz + csz));      // ... and started on its way
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
    }
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
    for (int k = 0; k < 2; ++k) vp8hip_host_free(0, buf[k]);
    printf("%s: %u frames %dx%d (coded %dx%d), %u key (%d by scene change, %d recoded), %zu bytes; %d hardware queues\n", argv[2], n, W, H, Wc, Hc, keys,
           st.scene_changes, st.redone_as_key, total, vp8hip_hw_queues());
    printf("%.6f s of reading + coding + writing (%.1f frames/s)\n", loop_s, n / (loop_s > 0 ? loop_s : 1));
    return 0;
}
