// pano_hostcopy.hpp - host side of the cv::Mat-shaped entry (pano_compose_host = ocvStitcher::process(vector<Mat>&, Mat&),
// reference include/ocvstitcher.hpp:1141): getting pageable caller memory to and from the GPU at PCIe rate.
//
// A hipMemcpy from pageable memory is staged by the runtime through a small bounce buffer on ONE thread: 4.5 GB/s measured
// on the MI355X box (62 panoramas/s for 8 x 1080p in, 2 x 3893 x 991 out).  Here the staging is explicit: a few copy threads
// move the caller's rows into page-locked buffers the ctx owns (host memcpy scales with threads until the memory
// controllers saturate), and each camera's DMA is queued the moment its rows are in place, so the copy of camera i+1
// overlaps the DMA of camera i.  Caller memory that is already page-locked (hipHostMalloc / hipHostRegister /
// pano_host_alloc) skips the staging and is DMA'd directly.
// This header is plain C++ (no HIP): tests/test_graphcut_host.py builds the pool under ThreadSanitizer.
#pragma once

#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace pano {

class CopyPool {
  public:
    // process-wide, created on first use and never destroyed (worker threads must not outlive a static's destructor)
    static CopyPool& instance() {
        static CopyPool* pool = new CopyPool();
        return *pool;
    }
    int threads() const { return nthreads_; }

    // A batch of row copies with completion per job: submit() queues a job's rows in chunks for the pool, wait(job) returns
    // when that job's rows are in place - and copies queued chunks itself while it waits, so the calling thread is one of the
    // workers and the pool's threads are woken once per batch, not once per job.
    struct Latch {  // `left` only changes under `m`: the last worker is done with the latch before a waiter can see 0
        int left = 0;
        std::mutex m;
        std::condition_variable cv;
    };
    void submit(Latch& latch, uint8_t* dst, size_t dpitch, const uint8_t* src, size_t spitch, size_t width, int rows) {
        if (rows <= 0 || width == 0) return;
        const size_t total = width * (size_t)rows;
        int parts = (int)std::min<size_t>((size_t)nthreads_, std::max<size_t>(1, total / kMinChunk));
        parts = std::max(1, std::min(parts, rows));
        const int per = (rows + parts - 1) / parts;
        {
            std::lock_guard<std::mutex> gl(latch.m);
            latch.left += parts;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            ensure_workers();
            for (int k = 0; k < parts; k++) {
                const int r0 = k * per, r1 = std::min(rows, r0 + per);
                q_.push_back(Task{dst + (size_t)r0 * dpitch, dpitch, src + (size_t)r0 * spitch, spitch, width, std::max(0, r1 - r0), &latch});
            }
        }
        cv_.notify_all();
    }
    void wait(Latch& latch) {
        for (;;) {
            {
                std::lock_guard<std::mutex> gl(latch.m);
                if (latch.left == 0) return;
            }
            Task t;
            bool have = false;
            {
                std::lock_guard<std::mutex> g(m_);
                if (!q_.empty()) {
                    t = q_.front();
                    q_.pop_front();
                    have = true;
                }
            }
            if (have) {
                run(t);
                continue;
            }
            std::unique_lock<std::mutex> lk(latch.m);   // nothing left to help with: the last chunks are in other hands
            latch.cv.wait(lk, [&] { return latch.left == 0; });
            return;
        }
    }
    // copy `rows` rows of `width` bytes, split over the pool and the calling thread; returns when done
    void copy2d(uint8_t* dst, size_t dpitch, const uint8_t* src, size_t spitch, size_t width, int rows) {
        if (rows <= 0 || width == 0) return;
        if (nthreads_ <= 1 || width * (size_t)rows < 2 * kMinChunk) {
            rows_copy(dst, dpitch, src, spitch, width, rows);
            return;
        }
        Latch latch;
        submit(latch, dst, dpitch, src, spitch, width, rows);
        wait(latch);
    }

  private:
    static constexpr size_t kMinChunk = 256 << 10;  // below this a task costs more than it copies
    struct Task {
        uint8_t* dst; size_t dpitch; const uint8_t* src; size_t spitch; size_t width; int rows; Latch* latch;
    };
    CopyPool() {
        int n = 8;  // PANO_HOST_THREADS: copy threads per process (1 = copy on the calling thread)
        if (const char* e = getenv("PANO_HOST_THREADS")) n = atoi(e);
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0) n = std::min(n, hw);
        nthreads_ = std::max(1, std::min(n, 64));
    }
    static void rows_copy(uint8_t* dst, size_t dpitch, const uint8_t* src, size_t spitch, size_t width, int rows) {
        if (dpitch == width && spitch == width) {
            std::memcpy(dst, src, width * (size_t)rows);
            return;
        }
        for (int y = 0; y < rows; y++) std::memcpy(dst + (size_t)y * dpitch, src + (size_t)y * spitch, width);
    }
    void ensure_workers() {  // m_ held
        if (pid_ != getpid()) {  // after a fork the child has this object but none of its threads
            workers_.clear();    // (detached: nothing to join)
            q_.clear();
            pid_ = getpid();
        }
        while ((int)workers_.size() < nthreads_ - 1) {
            workers_.emplace_back([this] { work(); });
            workers_.back().detach();
        }
    }
    void work() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return !q_.empty(); });
                t = q_.front();
                q_.pop_front();
            }
            run(t);
        }
    }
    static void run(const Task& t) {
        rows_copy(t.dst, t.dpitch, t.src, t.spitch, t.width, t.rows);
        std::lock_guard<std::mutex> g(t.latch->m);
        if (--t.latch->left == 0) t.latch->cv.notify_all();
    }
    int nthreads_ = 1;
    pid_t pid_ = 0;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Task> q_;
    std::vector<std::thread> workers_;
};

}  // namespace pano
