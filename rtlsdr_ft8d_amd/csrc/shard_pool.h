// shard_pool.h -- persistent host workers of the multi-GPU entries (api_multi.hip).  Plain C++17, no HIP: the same
// header is compiled with -fsanitize=thread by tests/host_tsan (tests/test_sanitizers.py) and stressed on the CPU.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <utility>

// Persistent host workers of the multi-GPU entries: one thread per concurrently running shard beyond the caller's
// own, created on first use and kept for the life of the process (round 2 spawned ndev-1 std::threads per call,
// which is measurable on small batches).  Workers carry no GPU state of their own: every task enters its context
// through the usual Entry guard.  No C++ exception crosses the C ABI: a failed thread creation makes post() return
// false and the caller runs the shard itself.
class ShardPool {
public:
    struct Latch {
        std::mutex m;
        std::condition_variable cv;
        int pending = 0;
        void wait() { std::unique_lock<std::mutex> l(m); cv.wait(l, [this] { return pending == 0; }); }
    };
    static ShardPool &instance() { static ShardPool *p = new ShardPool(); return *p; }    // never destroyed: no join at exit
    bool post(std::function<void()> fn, Latch *latch) {
        std::unique_lock<std::mutex> l(m_);
        try {
            q_.emplace_back(std::move(fn), latch);
        } catch (...) { return false; }
        // one waiting (or starting) worker per queued job, so that shards never queue up behind each other
        if (idle_ + starting_ < (int)q_.size()) {
            try { std::thread(&ShardPool::loop, this).detach(); ++workers_; ++starting_; }
            catch (...) {
                if (idle_ + starting_ == 0 && workers_ == 0) { q_.pop_back(); return false; }   // nobody would ever run it
            }
        }
        { std::lock_guard<std::mutex> g(latch->m); ++latch->pending; }
        l.unlock();
        cv_.notify_one();
        return true;
    }
    int workers() { std::lock_guard<std::mutex> l(m_); return workers_; }
private:
    void loop() {
        bool first = true;
        for (;;) {
            std::pair<std::function<void()>, Latch *> job;
            {
                std::unique_lock<std::mutex> l(m_);
                if (first) { --starting_; first = false; }
                ++idle_;
                cv_.wait(l, [this] { return !q_.empty(); });
                --idle_;
                job = std::move(q_.front());
                q_.pop_front();
            }
            job.first();
            // The latch lives on the caller's stack (run_shards): the waiter may return, and its frame may die, as soon as
            // it can observe pending == 0 -- which needs the mutex.  So the notification is sent while the mutex is still
            // held; after the unlock this thread never touches the latch again.
            {
                std::lock_guard<std::mutex> g(job.second->m);
                --job.second->pending;
                job.second->cv.notify_all();
            }
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::pair<std::function<void()>, Latch *>> q_;
    int workers_ = 0, idle_ = 0, starting_ = 0;
};
