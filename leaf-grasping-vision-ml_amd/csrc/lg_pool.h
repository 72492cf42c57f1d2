// Host worker pool of liblgrasp.so (plain C++17, no HIP: tests/test_host_pool.py builds it with -fsanitize=thread).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

// Persistent host worker pool (orientation + result post-processing): spawning 16 threads per call costs
// more than the work itself at B = 128.  A job is a heap object that the workers reference-count: a worker that was
// pre-empted between fetching an index and testing it can never see the NEXT job's bounds or function (the
// first version kept fn / n / next in the pool itself).
class LgPool {
    struct Job {
        const std::function<void(int)>* fn;
        int n;
        std::atomic<int> next{0}, done{0};
    };

public:
    explicit LgPool(int n) {
        for (int i = 0; i < n; i++) th_.emplace_back([this] { loop(); });
    }
    ~LgPool() {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int size() const { return (int)th_.size(); }
    void run(int n, const std::function<void(int)>& fn) {  // calls fn(0..n-1), returns when all are done
        if (n <= 0) return;
        if (n == 1 || th_.empty()) {
            for (int i = 0; i < n; i++) fn(i);
            return;
        }
        auto job = std::make_shared<Job>();
        job->fn = &fn;
        job->n = n;
        {
            std::lock_guard<std::mutex> l(m_);
            job_ = job;
            gen_++;
        }
        cv_.notify_all();
        work(*job);  // the caller helps
        std::unique_lock<std::mutex> l(m_);
        cv_done_.wait(l, [&] { return job->done.load() >= n; });
        if (job_ == job) job_.reset();   // fn dies with this frame; late workers only touch the counters of their copy
    }

private:
    void work(Job& j) {
        for (;;) {
            const int i = j.next.fetch_add(1);
            if (i >= j.n) break;
            (*j.fn)(i);      // i < n and not yet counted in done: run() is still waiting, fn is alive
            if (j.done.fetch_add(1) + 1 >= j.n) {
                std::lock_guard<std::mutex> l(m_);
                cv_done_.notify_all();
            }
        }
    }
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            std::shared_ptr<Job> j;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                j = job_;
            }
            if (j) work(*j);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, cv_done_;
    std::shared_ptr<Job> job_;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};

