// writer_bench: how fast does ONE file on this file system take the SAM text of a run?  (g++ -O2 -pthread tools/writer_bench.cpp -o /tmp/writer_bench)
// The SAM writer of the front end (star_amd/csrc/host/runner.cpp: writerLoop) appends ~230 MB per batch of 400 k pairs to Aligned.out.sam; on the GPU boxes
// the output directory is tmpfs and the writer, not the GPU, set the step of the pipeline.  This program replays the writer's access pattern without the rest:
// B batches of S bytes from R source buffers (the per-range text buffers of the formatting threads), by one of the methods below, W threads.
//   pwrite      W threads, each range at its offset (serialised by the inode lock)
//   mmap        grow the file by the batch, map the new part, W threads memcpy (the shipped writer of round 4)
//   falloc+mmap fallocate the batch's part first (pages handed out inside one system call), then as mmap
//   ahead       a helper thread keeps the file fallocate'd A batches ahead of the writer; W threads memcpy into the mapping (pages exist: minor faults only)
//   ahead+pw    as ahead, the copy by pwrite from W threads
//   populate    map, every thread madvise(MADV_POPULATE_WRITE) on its share, then memcpy
//   window      the file is grown and mapped 8 batches at a time (one mmap / munmap per 8 batches), W threads memcpy
// usage: writer_bench DIR [batches=25] [MB per batch=230] [threads=4] [busy=0]      busy = N other threads spinning (the formatting threads of the pipeline)
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Src { std::vector<std::vector<char> > r; std::vector<uint64_t> at; uint64_t total = 0; };

template <class F> static void onThreads(uint32_t W, F f) {
    std::vector<std::thread> th;
    for (uint32_t i = 1; i < W; i++) th.emplace_back(f, i);
    f(0);
    for (auto &x : th) x.join();
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: writer_bench DIR [batches] [MB] [threads] [busy]\n"); return 2; }
    const std::string dir = argv[1];
    const int B = argc > 2 ? atoi(argv[2]) : 25;
    const uint64_t S = (uint64_t)(argc > 3 ? atoi(argv[3]) : 230) << 20;
    const uint32_t W = argc > 4 ? (uint32_t)atoi(argv[4]) : 4;
    const int busy = argc > 5 ? atoi(argv[5]) : 0;
    const uint32_t R = 64;
    Src s; s.r.resize(R); s.at.assign(R + 1, 0);
    for (uint32_t i = 0; i < R; i++) { s.r[i].assign(S / R + (i * 977) % 4096, (char)('A' + i % 26)); s.at[i + 1] = s.at[i] + s.r[i].size(); }
    s.total = s.at[R];
    std::atomic<bool> stop(false);
    std::vector<std::thread> spin;
    for (int i = 0; i < busy; i++) spin.emplace_back([&] { volatile uint64_t x = 0; while (!stop) x += 1; });
    const char *methods[] = {"pwrite", "mmap", "falloc+mmap", "ahead", "ahead+pw", "populate", "window"};
    for (int rep = 0; rep < 2; rep++)
    for (const char *m : methods) {
        const std::string path = dir + "/writer_bench.out";
        unlink(path.c_str());
        int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) { perror("open"); return 1; }
        const std::string M = m;
        // the helper of "ahead": keeps [0, want) allocated
        std::mutex am; std::condition_variable acv; uint64_t want = 0, have = 0; bool astop = false;
        std::thread helper;
        const bool ahead = M == "ahead" || M == "ahead+pw";
        if (ahead) helper = std::thread([&] {
            for (;;) {
                uint64_t w;
                { std::unique_lock<std::mutex> l(am); acv.wait(l, [&] { return want > have || astop; }); if (astop) return; w = want; }
                if (fallocate(fd, FALLOC_FL_KEEP_SIZE, (off_t)have, (off_t)(w - have)) != 0) { perror("fallocate"); }
                { std::lock_guard<std::mutex> l(am); have = w; }
                acv.notify_all();
            }
        });
        uint64_t pos = 0; double worst = 0;
        char *win = nullptr; uint64_t winOff = 0, winLen = 0;
        if (ahead) { std::lock_guard<std::mutex> l(am); want = 3 * s.total; acv.notify_all(); }
        const double t0 = now();
        for (int b = 0; b < B; b++) {
            const double tb = now();
            const uint64_t end = pos + s.total;
            if (ahead) {
                { std::unique_lock<std::mutex> l(am); acv.wait(l, [&] { return have >= end; }); want = std::max(want, end + 3 * s.total); }
                acv.notify_all();
            }
            char *map = nullptr; uint64_t mapOff = 0, mapLen = 0;
            if (M == "window") {
                if (!win || end > winOff + winLen) {
                    if (win) munmap(win, winLen);
                    winOff = pos & ~4095ull; winLen = 8 * s.total + 4096;
                    if (ftruncate(fd, (off_t)(winOff + winLen)) != 0) { perror("ftruncate"); return 1; }
                    win = (char *)mmap(nullptr, winLen, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)winOff);
                    if (win == MAP_FAILED) { perror("mmap"); return 1; }
                }
                map = win; mapOff = winOff;
            } else if (M != "pwrite" && M != "ahead+pw") {
                if (ftruncate(fd, (off_t)end) != 0) { perror("ftruncate"); return 1; }
                if (M == "falloc+mmap" && fallocate(fd, 0, (off_t)pos, (off_t)s.total) != 0) perror("fallocate");
                mapOff = pos & ~4095ull; mapLen = end - mapOff;
                map = (char *)mmap(nullptr, mapLen, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)mapOff);
                if (map == MAP_FAILED) { perror("mmap"); return 1; }
                if (M == "populate") onThreads(W, [&](uint32_t i) { uint64_t per = ((mapLen / W) + 4095) & ~4095ull, lo = std::min(mapLen, per * i), hi = std::min(mapLen, lo + per); if (hi > lo) madvise(map + lo, hi - lo, MADV_POPULATE_WRITE); });
            }
            std::atomic<uint32_t> next(0);
            onThreads(W, [&](uint32_t) {
                for (;;) {
                    const uint32_t t = next.fetch_add(1);
                    if (t >= R) break;
                    const char *p = s.r[t].data(); uint64_t left = s.r[t].size(), off = pos + s.at[t];
                    if (map) { memcpy(map + (off - mapOff), p, left); continue; }
                    while (left) { ssize_t w = pwrite(fd, p, left, (off_t)off); if (w <= 0) { perror("pwrite"); exit(1); } p += w; left -= (uint64_t)w; off += (uint64_t)w; }
                }
            });
            if (map && map != win) munmap(map, mapLen);
            pos = end;
            worst = std::max(worst, now() - tb);
        }
        if (win) munmap(win, winLen);
        if (M == "window" || ahead) { if (ftruncate(fd, (off_t)pos) != 0) perror("ftruncate"); }
        const double dt = now() - t0;
        if (ahead) { { std::lock_guard<std::mutex> l(am); astop = true; } acv.notify_all(); helper.join(); }
        struct stat st; fstat(fd, &st);
        printf("%-12s W=%u busy=%d: %6.2f GB/s  %6.1f ms per batch (worst %6.1f)  file %.2f GB\n", m, W, busy, (double)pos / dt / 1e9, dt / B * 1e3, worst * 1e3, (double)st.st_size / 1e9);
        fflush(stdout);
        close(fd); unlink(path.c_str());
    }
    stop = true;
    for (auto &x : spin) x.join();
    return 0;
}
