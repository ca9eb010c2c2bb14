// gsplat_bench -- C++ caller of the C-ABI that mirrors the reference's main loop
// (Main.cpp:11-31 -> Engine::init, Engine/Engine.cpp:32-85): create, load the scene, set the
// scene's benchmark camera, warm up, then average the five GPU timing buckets the way
// Renderer.cpp:477-510 does, and print them.
//
//   gsplat_bench <scene.ply | --synthetic N> [--scene garden|train|bicycle|origin] [--res WxH]
//                [--warmup F] [--frames F] [--fast] [--sort radix4|splat_first|bucket|radix8|radix8_splat_first] [--out frame.png|frame.ppm]
//                [--ranks R [--interleaved]]
//
// --ranks R: the multi-GPU frame of SURVEY 8(e) without any Python -- R processes, one per GPU (rank r on device r),
// forked BEFORE anything touches a GPU; every rank loads the same scene, owns a band of tile rows (or every R-th row:
// --interleaved), renders it into a strip in HBM, and the strips meet on rank 0 through gs_gather_strips (RCCL,
// point-to-point over xGMI).  Rank 0 creates the communicator id and hands it to its siblings through pipes opened
// before the fork; it prints the mean frame time over the gather and writes the assembled frame with --out.
#include "../include/gsplat.h"

#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static uint64_t sm(uint64_t& s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float uni(uint64_t& s) { return (float)((sm(s) >> 40) * (1.0 / 16777216.0)); }

static bool write_all(int fd, const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n) { const ssize_t k = write(fd, c, n); if (k <= 0) return false; c += k; n -= (size_t)k; }
    return true;
}
static bool read_all(int fd, void* p, size_t n) {
    char* c = static_cast<char*>(p);
    while (n) { const ssize_t k = read(fd, c, n); if (k <= 0) return false; c += k; n -= (size_t)k; }
    return true;
}

// One rank of a sharded frame (after the fork; `id_fd`: read end of this rank's pipe, or the write ends on rank 0).
static int run_rank(gs_ctx* ctx, int rank, int ranks, bool interleaved, const std::vector<int>& id_fds, uint32_t w, uint32_t h,
                    const float* view, const float* proj, const float* pos, uint32_t warmup, uint32_t frames, const std::string& out) {
    unsigned char id[GS_DIST_UNIQUE_ID_BYTES];
    if (rank == 0) {
        if (gs_dist_unique_id(id) != GS_OK) { fprintf(stderr, "[Log Error]: gs_dist_unique_id failed\n"); return 1; }
        for (int fd : id_fds) if (!write_all(fd, id, sizeof(id))) { fprintf(stderr, "[Log Error]: cannot hand the id to a rank\n"); return 1; }
    } else if (!read_all(id_fds[0], id, sizeof(id))) { fprintf(stderr, "[Log Error]: rank %d got no id\n", rank); return 1; }
    if (gs_dist_init(ctx, id, rank, ranks) != GS_OK) { fprintf(stderr, "[Log Error]: rank %d: %s\n", rank, gs_last_error(ctx)); return 1; }
    if (gs_dist_shard_rows(ctx, interleaved ? 1u : 0u) != GS_OK) { fprintf(stderr, "[Log Error]: rank %d: %s\n", rank, gs_last_error(ctx)); return 1; }
    std::vector<uint8_t> img(rank == 0 ? (size_t)w * h * 4 : 0);
    double ms = 0.0;
    for (uint32_t f = 0; f < warmup + frames; ++f) {         // every rank draws every frame (Engine.cpp:45-78)
        const auto t0 = std::chrono::steady_clock::now();
        if (gs_render_sharded(ctx, view, proj, pos, 0, rank == 0 ? img.data() : nullptr) < 0) {
            fprintf(stderr, "[Log Error]: rank %d: %s\n", rank, gs_last_error(ctx)); return 1;
        }
        if (f >= warmup) ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (rank == 0) {
        printf("ranks: %d (%s tile rows)   frame + gather + copy to host ms (host clock, rank 0): %.3f\n", ranks,
               interleaved ? "interleaved" : "contiguous", ms / std::max(1u, frames));
        if (!out.empty() && gs_write_image(out.c_str(), img.data(), w, h) != GS_OK) fprintf(stderr, "cannot write %s\n", out.c_str());
    }
    return 0;
}

int main(int argc, char** argv) {
    std::string ply, scene = "origin", ppm, sort = "radix4";   // GPU_SORT_ALGORITHM (Renderer.h:33)
    uint32_t n_syn = 0, w = 1280, h = 720, warmup = 1000, frames = 1000;   // window 1280x720: Engine.cpp:35; WAIT_ELAPSED_*_FRAMES_FOR_AVG: Renderer.h:142-143
    bool fast = false, interleaved = false;
    int ranks = 1, rank = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--synthetic" && i + 1 < argc) n_syn = (uint32_t)atol(argv[++i]);
        else if (a == "--scene" && i + 1 < argc) scene = argv[++i];
        else if (a == "--res" && i + 1 < argc) sscanf(argv[++i], "%ux%u", &w, &h);
        else if (a == "--warmup" && i + 1 < argc) warmup = (uint32_t)atol(argv[++i]);
        else if (a == "--frames" && i + 1 < argc) frames = (uint32_t)atol(argv[++i]);
        else if ((a == "--out" || a == "--ppm") && i + 1 < argc) ppm = argv[++i];
        else if (a == "--fast") fast = true;
        else if (a == "--sort" && i + 1 < argc) sort = argv[++i];
        else if (a == "--ranks" && i + 1 < argc) ranks = atoi(argv[++i]);
        else if (a == "--interleaved") interleaved = true;
        else ply = a;
    }
    if (ply.empty() && !n_syn) { fprintf(stderr, "usage: gsplat_bench <scene.ply | --synthetic N> [...]\n"); return 2; }
    if (ranks < 1 || ranks > 64) { fprintf(stderr, "--ranks must be in [1, 64]\n"); return 2; }

    // one process per GPU, forked before any GPU call (nothing above this line touches HIP); rank 0 stays the parent
    std::vector<int> id_fds;                 // rank 0: write ends towards ranks 1 .. R-1; rank r: its read end
    std::vector<pid_t> children;
    // GSPLAT_BENCH_FORK_TEST (tests only): R processes that all render the WHOLE frame on device 0, no communicator --
    // what a one-GPU box can check of "fork first, touch the GPU afterwards"
    const bool fork_test = std::getenv("GSPLAT_BENCH_FORK_TEST") != nullptr;
    const bool sharded = !fork_test && (ranks > 1 || interleaved || std::getenv("GSPLAT_BENCH_DIST") != nullptr);   // the last two: the R = 1 form of the path
    if (ranks > 1) {
        std::vector<int> wr;
        for (int r = 1; r < ranks && rank == 0; ++r) {
            int fd[2];
            if (pipe(fd) != 0) { perror("pipe"); return 1; }
            const pid_t pid = fork();
            if (pid < 0) { perror("fork"); return 1; }
            if (pid == 0) { rank = r; close(fd[1]); for (int o : wr) close(o); id_fds.assign(1, fd[0]); children.clear(); }
            else { close(fd[0]); wr.push_back(fd[1]); children.push_back(pid); }
        }
        if (rank == 0) id_fds = wr;
    }

    gs_config cfg; gs_default_config(&cfg);
    // GSPLAT_BENCH_SAME_DEVICE (tests only, with tools/mock_rccl first on LD_LIBRARY_PATH: RCCL itself refuses two ranks on
    // one device): every rank on device 0, the real sharded path
    cfg.device_ordinal = fork_test || std::getenv("GSPLAT_BENCH_SAME_DEVICE") ? 0 : rank;
    cfg.render_mode = fast ? GS_RENDER_FAST : GS_RENDER_EXACT;
    cfg.sort_algorithm = sort == "splat_first" ? GS_SORT_RADIX4_SPLAT_FIRST : sort == "bucket" ? GS_SORT_TILE_BUCKET
                       : sort == "radix8" ? GS_SORT_RADIX8 : sort == "radix8_splat_first" ? GS_SORT_RADIX8_SPLAT_FIRST : GS_SORT_RADIX4;
    cfg.record_timings = 1;                                  // RECORD_GPU_TIMES (GfxSettings.h:7) on: this is the benchmark build
    gs_ctx* ctx = nullptr;
    if (gs_create(&cfg, &ctx) != GS_OK) { fprintf(stderr, "[Log Error]: %s\n", gs_last_error(nullptr)); return 1; }

    if (!ply.empty()) {                                     // Scene::init -> ResourceManager::loadGaussians
        int rc = gs_load_ply(ctx, ply.c_str());
        if (rc != GS_OK) { fprintf(stderr, "[Log Error]: %s (%d)\n", gs_ply_last_error(), rc); return 1; }
    } else {                                                // simple cloud in front of an origin camera
        std::vector<float> rec((size_t)n_syn * 84, 0.0f);
        uint64_t s = 20240807;
        const float aspect = (float)w / (float)h;
        for (uint32_t i = 0; i < n_syn; ++i) {
            float* g = &rec[(size_t)i * 84];
            const float d = 0.5f + 19.5f * uni(s);
            g[0] = d * aspect * (-1.5f + 3.0f * uni(s)); g[1] = d * (-1.5f + 3.0f * uni(s)); g[2] = d;
            for (int a = 0; a < 3; ++a) g[4 + a] = std::exp(-4.0f + 1.2f * (uni(s) - 0.5f));
            float q[4], l = 0; for (int a = 0; a < 4; ++a) { q[a] = uni(s) - 0.5f; l += q[a] * q[a]; }
            l = 1.0f / std::sqrt(l + 1e-12f); for (int a = 0; a < 4; ++a) g[8 + a] = q[a] * l;
            for (int c = 0; c < 3; ++c) g[12 + c] = -1.5f + 3.0f * uni(s);
            g[15] = 1.0f / (1.0f + std::exp(2.0f - 6.0f * uni(s)));
            for (int k = 16; k < 76; ++k) if ((k & 3) != 3) g[k] = 0.1f * (uni(s) - 0.5f);
        }
        if (gs_upload_gaussians(ctx, rec.data(), n_syn) != GS_OK) { fprintf(stderr, "[Log Error]: %s\n", gs_last_error(ctx)); return 1; }
    }
    if (gs_set_resolution(ctx, w, h) != GS_OK) { fprintf(stderr, "[Log Error]: %s\n", gs_last_error(ctx)); return 1; }

    // "Camera for benchmarks" poses: GardenScene.cpp:11-12, TrainScene.cpp:11-12, BicycleScene.cpp:11-12
    float pos[3] = {0, 0, 0}, yaw = 0, pitch = 0;
    if (scene == "garden") { pos[0] = -0.620010f; pos[1] = 0.189628f; pos[2] = 2.271181f; yaw = 2.971590f; pitch = -1.074159f; }
    else if (scene == "train") { pos[0] = -2.857887f; pos[1] = 0.188856f; pos[2] = 1.048745f; yaw = 1.361593f; pitch = 0.005841f; }
    else if (scene == "bicycle") { pos[0] = 0.945927f; pos[1] = -0.294418f; pos[2] = -0.181088f; yaw = -1.108407f; pitch = -0.324159f; }
    float view[16], proj[16];
    gs_camera_matrices(pos, yaw, pitch, (float)w / (float)h, cfg.near_plane, cfg.far_plane, view, proj);

    if (sharded) {
        int rc = run_rank(ctx, rank, ranks, interleaved, id_fds, w, h, view, proj, pos, warmup, frames, ppm);
        gs_destroy(ctx);
        for (int fd : id_fds) close(fd);
        for (pid_t pid : children) { int st = 0; waitpid(pid, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 1; }
        return rc;
    }

    gs_scene_info info; gs_get_scene_info(ctx, &info);
    printf("[Log]: Number of gaussians: %u   sort list capacity: %u   passes: %u\n", info.num_gaussians, info.capacity, info.num_sort_bits / 4);
    double avg[5] = {0, 0, 0, 0, 0};
    double havg[4] = {0, 0, 0, 0};                           // RECORD_CPU_TIMES averages (Renderer.cpp:430-433)
    gs_timings t{};
    for (uint32_t f = 0; f < warmup + frames; ++f) {        // Engine.cpp:45-78 / Renderer.cpp:477-488
        int rc = gs_render_device(ctx, view, proj, pos, 0, nullptr);
        if (rc < 0) { fprintf(stderr, "[Log Error]: %s\n", gs_last_error(ctx)); return 1; }
        gs_get_timings(ctx, &t);
        if (f >= warmup) {
            const double k = 1.0 / (double)(f - warmup + 1);
            const double v[5] = {t.init_sort_list_ms, t.radix_sort_ms, t.find_ranges_ms, t.render_ms, t.total_ms};
            for (int b = 0; b < 5; ++b) avg[b] = (1.0 - k) * avg[b] + k * v[b];
            gs_host_timings ht{};
            gs_get_host_timings(ctx, &ht);
            const double hv[4] = {ht.wait_ms, ht.record_ms, ht.present_ms, ht.cpu_frame_ms};
            for (int b = 0; b < 4; ++b) havg[b] = (1.0 - k) * havg[b] + k * hv[b];
        }
    }
    printf("elements to sort: %u%s\n", t.num_sort_elements, t.overflowed ? " (overflowed, truncated)" : "");
    printf("init sort list ms: %.3f\nsort ms: %.3f\nfind ranges ms: %.3f\nrender gaussians ms: %.3f\ntotal gpu time ms: %.3f\n",
           avg[0], avg[1], avg[2], avg[3], avg[4]);
    printf("Msplats/s: %.1f\n", info.num_gaussians / avg[4] / 1000.0);
    printf("waitForFence ms: %.4f\nrecordCommandBuffer ms: %.4f\npresent ms: %.4f\nCPU frame time ms: %.4f\n",
           havg[0], havg[1], havg[2], havg[3]);
    if (fork_test && !ppm.empty() && rank > 0) {                                          // every process its own file: frame.ppm -> frame.1.ppm
        const size_t dot = ppm.rfind('.');
        ppm.insert(dot == std::string::npos ? ppm.size() : dot, "." + std::to_string(rank));
    }
    if (!ppm.empty()) {
        std::vector<uint8_t> img((size_t)w * h * 4);
        gs_debug_read(ctx, GS_BUF_IMAGE, img.data(), img.size());
        if (gs_write_image(ppm.c_str(), img.data(), w, h) != GS_OK)    // .ppm or .png by extension
            fprintf(stderr, "cannot write %s\n", ppm.c_str());
    }
    gs_destroy(ctx);
    int rc_children = 0;
    for (pid_t pid : children) { int st = 0; waitpid(pid, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc_children = 1; }
    return rc_children;
}
