// gsplat_bench -- the C++ host a maintainer of the reference would write: the reference's main loop (Main.cpp:11-31 ->
// Engine::init, Engine/Engine.cpp:32-85) over gsplat::Renderer (include/gsplat.hpp), the class INTEGRATION.md swaps in for
// the reference's Renderer: init, initForScene / initForScenePly, the scene's benchmark camera, draw every frame, and the
// running averages of the five GPU timing buckets (Renderer.cpp:477-510) printed at the end.
//
//   gsplat_bench <scene.ply | --synthetic N [--skew]> [--scene garden|train|bicycle|origin] [--res WxH]
//                [--warmup F] [--frames F] [--fast] [--sort radix4|splat_first|bucket|radix8|radix8_splat_first]
//                [--present] [--out frame.png|frame.ppm]
//                [--ranks R [--interleaved | --balanced [--rebalance K]] [--sync]]
//
// --skew: the synthetic cloud crowds towards the top of the frame (what a sky-less capture does to the upper tile rows), so
// that equal bands are NOT equal work and --balanced has something to move.
// --present: draw() copies every frame to the host (the windowless sink that stands in for the swapchain present); without
// it drawDevice() leaves the frame in HBM, as the reference's frame stays in the swapchain image.
// --ranks R: the multi-GPU frame of SURVEY 8(e) without any Python -- R processes, one per GPU (rank r on device r),
// forked BEFORE anything touches a GPU; every rank loads the same scene and owns tile rows (a band of equal height;
// --interleaved: every R-th row; --balanced: bands whose edges follow the scene, re-cut every K frames by rebalance()).
// Default: drawShardedAsync -- two frames in flight, the exchange (RCCL, point-to-point over xGMI) beside the next frame's
// kernels, the assembled frame left in rank 0's HBM; the mean frame time is printed without a host copy and, measured
// again, with one copy per frame (shardedRead of the frame before).  --sync: drawSharded (frame + exchange + copy, nothing
// overlaps).  Rank 0 creates the communicator id and hands it to its siblings through pipes opened before the fork; a rank
// that fails before the communicator exists says so over its pipe, and everybody leaves instead of waiting in
// ncclCommInitRank for ever.
#include "../include/gsplat.hpp"

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

// The ranks' side channel: pipes opened before the fork.  Rank 0 holds, per sibling, a write end (down) and a read end (up).
struct Channel {
    int rank = 0, ranks = 1;
    std::vector<int> down, up;       // rank 0: [r - 1] towards / from rank r; rank r: one element each
    // every rank says whether it is ready (a dead sibling reads as "not"); rank 0 answers everybody with go (1) / leave (0)
    // and `payload` (the communicator id, first round only).  Returns true when all ranks go on.
    bool agree(bool mine, void* payload, size_t bytes) {
        if (ranks == 1) return mine;
        if (rank == 0) {
            bool all = mine;
            for (int fd : up) { char ok = 0; if (!read_all(fd, &ok, 1) || !ok) all = false; }
            const char go = all ? 1 : 0;
            for (int fd : down) { (void)write_all(fd, &go, 1); if (all && bytes) (void)write_all(fd, payload, bytes); }
            return all;
        }
        const char ok = mine ? 1 : 0;
        char go = 0;
        if (!write_all(up[0], &ok, 1) || !read_all(down[0], &go, 1) || !go) return false;
        return bytes == 0 || read_all(down[0], payload, bytes);
    }
};

static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

struct Options {
    uint32_t w = 1280, h = 720, warmup = 1000, frames = 1000;   // window 1280x720: Engine.cpp:35; WAIT_ELAPSED_*_FRAMES_FOR_AVG: Renderer.h:142-143
    uint32_t dealing = GS_ROWS_CONTIGUOUS, rebalance_every = 0;
    bool sync = false;
    std::string out;
};

// One rank of a sharded frame.
static int run_rank(gsplat::Renderer& r, Channel& ch, const Options& o, const float* view, const float* proj, const float* pos) {
    unsigned char id[GS_DIST_UNIQUE_ID_BYTES] = {};
    bool ok = true;
    if (ch.rank == 0 && gsplat::Renderer::distUniqueId(id) != GS_OK) { fprintf(stderr, "[Log Error]: %s\n", gs_last_error(nullptr)); ok = false; }
    // round 1: every rank has a context with the scene in it (or says it has not) -- only then does anybody enter ncclCommInitRank
    if (!ch.agree(ok, id, sizeof(id))) { fprintf(stderr, "[Log Error]: rank %d: a rank is not ready, leaving before the communicator\n", ch.rank); return 1; }
    ok = r.initDist(id, ch.rank, ch.ranks) == GS_OK;
    if (!ok) fprintf(stderr, "[Log Error]: rank %d: %s\n", ch.rank, r.lastError().c_str());
    if (ok && r.shardRows(o.dealing) != GS_OK) { fprintf(stderr, "[Log Error]: rank %d: %s\n", ch.rank, r.lastError().c_str()); ok = false; }
    // round 2: a rank without its buffers would leave the others waiting in the first exchange (gsplat.h, "Collective safety")
    if (!ch.agree(ok, nullptr, 0)) { fprintf(stderr, "[Log Error]: rank %d: a rank could not set up its rows, leaving\n", ch.rank); return 1; }
    const bool root = ch.rank == 0;
    std::vector<uint8_t> img(root ? (size_t)o.w * o.h * 4 : 0);
    const char* how = o.dealing == GS_ROWS_INTERLEAVED ? "interleaved" : o.dealing == GS_ROWS_BALANCED ? "balanced" : "contiguous";
    uint32_t moves = 0;
    auto maybe_rebalance = [&](uint32_t f) {
        if (o.dealing != GS_ROWS_BALANCED || !o.rebalance_every || f == 0 || f % o.rebalance_every) return true;
        bool moved = false;
        if (r.rebalance(&moved) < 0) { fprintf(stderr, "[Log Error]: rank %d: %s\n", ch.rank, r.lastError().c_str()); return false; }
        moves += moved ? 1u : 0u;
        return true;
    };
    if (o.sync) {
        double ms = 0.0;
        for (uint32_t f = 0; f < o.warmup + o.frames; ++f) {         // every rank draws every frame (Engine.cpp:45-78)
            if (!maybe_rebalance(f)) return 1;
            const auto t0 = std::chrono::steady_clock::now();
            if (r.drawSharded(view, proj, pos, 0, root ? img.data() : nullptr) < 0) { fprintf(stderr, "[Log Error]: rank %d: %s\n", ch.rank, r.lastError().c_str()); return 1; }
            if (f >= o.warmup) ms += ms_since(t0);
        }
        if (root) printf("ranks: %d (%s tile rows)   frame + gather + copy to host ms (host clock, rank 0): %.3f\n", ch.ranks, how, ms / std::max(1u, o.frames));
        // every rank's own share (gsplat::Renderer::averages(): the GPU time of ITS rows, no exchange): a sharded frame ends with its
        // slowest rank, so rank 0 prints them all
        double mine = r.averages()[4];
        if (root) {
            std::vector<double> share(1, mine);
            for (int fd : ch.up) { double v = 0.0; if (!read_all(fd, &v, sizeof(v))) v = -1.0; share.push_back(v); }
            double worst = 0.0, sum = 0.0;
            printf("per-rank share, total gpu time ms:");
            for (double v : share) { printf(" %.3f", v); worst = std::max(worst, v); sum += v; }
            printf("   slowest / mean: %.3f\n", sum > 0.0 ? worst / (sum / (double)share.size()) : 0.0);
        } else if (!write_all(ch.up[0], &mine, sizeof(mine))) {
            fprintf(stderr, "[Log Error]: rank %d: cannot report its share time\n", ch.rank);
        }
    } else {
        // two frames in flight; pass 0: the frame stays in HBM, pass 1: the frame before is copied to the host every frame
        double ms[2] = {0.0, 0.0};
        for (int pass = 0; pass < 2; ++pass) {
            std::chrono::steady_clock::time_point t0;
            for (uint32_t f = 0; f < o.warmup + o.frames; ++f) {
                if (pass == 0 && !maybe_rebalance(f)) return 1;
                if (f == o.warmup) { void* d = nullptr; if (f) (void)r.shardedFrame(0, &d); t0 = std::chrono::steady_clock::now(); }
                if (r.drawShardedAsync(view, proj, pos, 0) < 0) { fprintf(stderr, "[Log Error]: rank %d: %s\n", ch.rank, r.lastError().c_str()); return 1; }
                if (pass == 1 && f > 0 && r.shardedRead(1, root ? img.data() : nullptr) < 0) { fprintf(stderr, "[Log Error]: rank %d: %s\n", ch.rank, r.lastError().c_str()); return 1; }
            }
            void* dev = nullptr;
            if (r.shardedFrame(0, &dev) < 0) { fprintf(stderr, "[Log Error]: rank %d: %s\n", ch.rank, r.lastError().c_str()); return 1; }
            ms[pass] = ms_since(t0) / std::max(1u, o.frames);
        }
        if (root) {
            printf("ranks: %d (%s tile rows)   frame + gather ms, two frames in flight, frame left in HBM (host clock, rank 0): %.3f\n", ch.ranks, how, ms[0]);
            printf("the same with the frame before copied to the host every frame: %.3f\n", ms[1]);
            if (r.shardedRead(0, img.data()) < 0) { fprintf(stderr, "[Log Error]: %s\n", r.lastError().c_str()); return 1; }
        }
    }
    if (root && o.dealing == GS_ROWS_BALANCED) {
        std::vector<uint32_t> edges((size_t)ch.ranks + 1u);
        if (gs_dist_bands(r.handle(), edges.data(), (uint32_t)edges.size()) == GS_OK) {
            printf("bands after %u moves:", moves);
            for (int k = 0; k < ch.ranks; ++k) printf(" %u-%u", edges[(size_t)k], edges[(size_t)k + 1u]);
            printf("\n");
        }
    }
    if (root && !o.out.empty() && gs_write_image(o.out.c_str(), img.data(), o.w, o.h) != GS_OK) fprintf(stderr, "cannot write %s\n", o.out.c_str());
    return 0;
}

int main(int argc, char** argv) {
    std::string ply, scene = "origin", sort = "radix4";   // GPU_SORT_ALGORITHM (Renderer.h:33)
    Options o;
    uint32_t n_syn = 0;
    bool fast = false, present = false, skew = false;
    int ranks = 1, rank = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--synthetic" && i + 1 < argc) n_syn = (uint32_t)atol(argv[++i]);
        else if (a == "--scene" && i + 1 < argc) scene = argv[++i];
        else if (a == "--res" && i + 1 < argc) sscanf(argv[++i], "%ux%u", &o.w, &o.h);
        else if (a == "--warmup" && i + 1 < argc) o.warmup = (uint32_t)atol(argv[++i]);
        else if (a == "--frames" && i + 1 < argc) o.frames = (uint32_t)atol(argv[++i]);
        else if ((a == "--out" || a == "--ppm") && i + 1 < argc) o.out = argv[++i];
        else if (a == "--fast") fast = true;
        else if (a == "--present") present = true;
        else if (a == "--skew") skew = true;
        else if (a == "--sort" && i + 1 < argc) sort = argv[++i];
        else if (a == "--ranks" && i + 1 < argc) ranks = atoi(argv[++i]);
        else if (a == "--interleaved") o.dealing = GS_ROWS_INTERLEAVED;
        else if (a == "--balanced") o.dealing = GS_ROWS_BALANCED;
        else if (a == "--rebalance" && i + 1 < argc) o.rebalance_every = (uint32_t)atol(argv[++i]);
        else if (a == "--sync") o.sync = true;
        else ply = a;
    }
    if (ply.empty() && !n_syn) { fprintf(stderr, "usage: gsplat_bench <scene.ply | --synthetic N> [...]\n"); return 2; }
    if (ranks < 1 || ranks > 64) { fprintf(stderr, "--ranks must be in [1, 64]\n"); return 2; }
    if (o.dealing == GS_ROWS_BALANCED && !o.rebalance_every) o.rebalance_every = 8;

    // one process per GPU, forked before any GPU call (nothing above this line touches HIP); rank 0 stays the parent
    Channel ch;
    std::vector<pid_t> children;
    // GSPLAT_BENCH_FORK_TEST (tests only): R processes that all render the WHOLE frame on device 0, no communicator --
    // what a one-GPU box can check of "fork first, touch the GPU afterwards"
    const bool fork_test = std::getenv("GSPLAT_BENCH_FORK_TEST") != nullptr;
    const bool sharded = !fork_test && (ranks > 1 || o.dealing != GS_ROWS_CONTIGUOUS || std::getenv("GSPLAT_BENCH_DIST") != nullptr);   // the last two: the R = 1 form of the path
    if (ranks > 1) {
        std::vector<int> down, up;
        for (int r = 1; r < ranks && rank == 0; ++r) {
            int d[2], u[2];
            if (pipe(d) != 0 || pipe(u) != 0) { perror("pipe"); return 1; }
            const pid_t pid = fork();
            if (pid < 0) { perror("fork"); return 1; }
            if (pid == 0) {
                rank = r; close(d[1]); close(u[0]);
                for (int fd : down) close(fd);
                for (int fd : up) close(fd);
                down.assign(1, d[0]); up.assign(1, u[1]); children.clear();
            } else { close(d[0]); close(u[1]); down.push_back(d[1]); up.push_back(u[0]); children.push_back(pid); }
        }
        ch.down = down; ch.up = up;
    }
    ch.rank = rank; ch.ranks = ranks;
    auto reap = [&](int rc) {                                   // rank 0: a sibling that failed fails the run
        for (int fd : ch.down) close(fd);
        for (int fd : ch.up) close(fd);
        for (pid_t pid : children) { int st = 0; waitpid(pid, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 1; }
        return rc;
    };

    gs_config cfg; gs_default_config(&cfg);
    // GSPLAT_BENCH_SAME_DEVICE (tests only, with tools/mock_rccl first on LD_LIBRARY_PATH: RCCL itself refuses two ranks on
    // one device): every rank on device 0, the real sharded path.  GSPLAT_BENCH_FAIL_RANK (tests only): that rank "has no GPU".
    cfg.device_ordinal = fork_test || std::getenv("GSPLAT_BENCH_SAME_DEVICE") ? 0 : rank;
    if (const char* fr = std::getenv("GSPLAT_BENCH_FAIL_RANK")) if (atoi(fr) == rank) cfg.device_ordinal = 1 << 20;
    cfg.render_mode = fast ? GS_RENDER_FAST : GS_RENDER_EXACT;
    cfg.sort_algorithm = sort == "splat_first" ? GS_SORT_RADIX4_SPLAT_FIRST : sort == "bucket" ? GS_SORT_TILE_BUCKET
                       : sort == "radix8" ? GS_SORT_RADIX8 : sort == "radix8_splat_first" ? GS_SORT_RADIX8_SPLAT_FIRST : GS_SORT_RADIX4;
    cfg.record_timings = 1;                                  // RECORD_GPU_TIMES (GfxSettings.h:7) on: this is the benchmark build

    gsplat::Renderer renderer(o.w, o.h, o.warmup, o.frames);    // Renderer.h:142-143, here from the command line
    bool ready = renderer.init(&cfg) == GS_OK;
    if (!ready) fprintf(stderr, "[Log Error]: rank %d: %s\n", rank, renderer.lastError().c_str());
    if (ready && !ply.empty()) {                            // Scene::init -> ResourceManager::loadGaussians
        ready = renderer.initForScenePly(ply) == GS_OK;
        if (!ready) fprintf(stderr, "[Log Error]: %s\n", renderer.lastError().c_str());
    } else if (ready) {                                     // simple cloud in front of an origin camera
        std::vector<float> rec((size_t)n_syn * 84, 0.0f);
        uint64_t s = 20240807;
        const float aspect = (float)o.w / (float)o.h;
        for (uint32_t i = 0; i < n_syn; ++i) {
            float* g = &rec[(size_t)i * 84];
            const float d = 0.5f + 19.5f * uni(s);
            g[0] = d * aspect * (-1.5f + 3.0f * uni(s));
            const float uy = uni(s);
            g[1] = d * (skew ? 1.2f - 2.7f * uy * uy * uy : -1.5f + 3.0f * uy); g[2] = d;     // world +y is up: the upper rows of the frame
            for (int a = 0; a < 3; ++a) g[4 + a] = std::exp(-4.0f + 1.2f * (uni(s) - 0.5f));
            float q[4], l = 0; for (int a = 0; a < 4; ++a) { q[a] = uni(s) - 0.5f; l += q[a] * q[a]; }
            l = 1.0f / std::sqrt(l + 1e-12f); for (int a = 0; a < 4; ++a) g[8 + a] = q[a] * l;
            for (int c = 0; c < 3; ++c) g[12 + c] = -1.5f + 3.0f * uni(s);
            g[15] = 1.0f / (1.0f + std::exp(2.0f - 6.0f * uni(s)));
            for (int k = 16; k < 76; ++k) if ((k & 3) != 3) g[k] = 0.1f * (uni(s) - 0.5f);
        }
        ready = renderer.initForScene(rec.data(), n_syn) == GS_OK;
        if (!ready) fprintf(stderr, "[Log Error]: %s\n", renderer.lastError().c_str());
    }

    // "Camera for benchmarks" poses: GardenScene.cpp:11-12, TrainScene.cpp:11-12, BicycleScene.cpp:11-12
    float pos[3] = {0, 0, 0}, yaw = 0, pitch = 0;
    if (scene == "garden") { pos[0] = -0.620010f; pos[1] = 0.189628f; pos[2] = 2.271181f; yaw = 2.971590f; pitch = -1.074159f; }
    else if (scene == "train") { pos[0] = -2.857887f; pos[1] = 0.188856f; pos[2] = 1.048745f; yaw = 1.361593f; pitch = 0.005841f; }
    else if (scene == "bicycle") { pos[0] = 0.945927f; pos[1] = -0.294418f; pos[2] = -0.181088f; yaw = -1.108407f; pitch = -0.324159f; }
    float view[16], proj[16];
    gs_camera_matrices(pos, yaw, pitch, (float)o.w / (float)o.h, cfg.near_plane, cfg.far_plane, view, proj);

    if (sharded) {
        // a rank that is not ready still meets the others in run_rank's first round, so that nobody waits for it
        int rc = 1;
        if (ready) rc = run_rank(renderer, ch, o, view, proj, pos);
        else (void)ch.agree(false, nullptr, 0);
        renderer.cleanup();
        return reap(rc);
    }
    if (!ready) return reap(1);

    gs_scene_info info; gs_get_scene_info(renderer.handle(), &info);
    printf("[Log]: Number of gaussians: %u   sort list capacity: %u   passes: %u\n", info.num_gaussians, info.capacity, info.num_sort_bits / 4);
    std::vector<uint8_t> img(present || !o.out.empty() ? (size_t)o.w * o.h * 4 : 0);
    std::vector<double> totals;                              // the last `frames` total_ms, to check the wrapper's running mean
    for (uint32_t f = 0; f < o.warmup + o.frames; ++f) {    // Engine.cpp:45-78: update, draw
        const int rc = present ? renderer.draw(view, proj, pos, 0, img.data()) : renderer.drawDevice(view, proj, pos, 0);
        if (rc < 0) { fprintf(stderr, "[Log Error]: %s\n", renderer.lastError().c_str()); return reap(1); }
        if (f >= o.warmup) totals.push_back(renderer.lastTimings().total_ms);
    }
    const gs_timings& t = renderer.lastTimings();
    const double* avg = renderer.averages();
    const double* havg = renderer.hostAverages();
    printf("elements to sort: %u%s\n", t.num_sort_elements, t.overflowed ? " (overflowed, truncated)" : "");
    printf("init sort list ms: %.3f\nsort ms: %.3f\nfind ranges ms: %.3f\nrender gaussians ms: %.3f\ntotal gpu time ms: %.3f\n",
           avg[0], avg[1], avg[2], avg[3], avg[4]);
    printf("Msplats/s: %.1f\n", info.num_gaussians / avg[4] / 1000.0);
    printf("waitForFence ms: %.4f\nrecordCommandBuffer ms: %.4f\npresent ms: %.4f\nCPU frame time ms: %.4f\n",
           havg[0], havg[1], havg[2], havg[3]);
    // Renderer.cpp:477-488 is a running mean: after warm-up + F frames it must be the plain mean of the last F frames
    double mean = 0.0;
    for (double v : totals) mean += v;
    mean /= std::max<size_t>(1, totals.size());
    printf("averages check: total gpu time running mean %.6f, mean of the last %zu frames %.6f, complete %d: %s\n", avg[4], totals.size(), mean,
           renderer.averagesComplete() ? 1 : 0,
           std::fabs(avg[4] - mean) <= 1e-9 + 1e-6 * std::fabs(mean) && renderer.averagesComplete() && renderer.elapsedFrames() == (uint64_t)o.warmup + o.frames ? "ok" : "MISMATCH");
    if (fork_test && !o.out.empty() && rank > 0) {                                          // every process its own file: frame.ppm -> frame.1.ppm
        const size_t dot = o.out.rfind('.');
        o.out.insert(dot == std::string::npos ? o.out.size() : dot, "." + std::to_string(rank));
    }
    if (!o.out.empty()) {
        if (!present) gs_debug_read(renderer.handle(), GS_BUF_IMAGE, img.data(), img.size());
        if (gs_write_image(o.out.c_str(), img.data(), o.w, o.h) != GS_OK)    // .ppm or .png by extension
            fprintf(stderr, "cannot write %s\n", o.out.c_str());
    }
    renderer.cleanup();
    return reap(0);
}
