// A small pool of worker threads for the host-side scene preparation (hierarchy build, packing, record set-up).
// One parallel region runs at a time; a region is a number of chunks handed out by an atomic counter, the calling thread
// takes chunks as well.  Chunk boundaries never depend on the number of threads, so results built from per-chunk partial
// results (combined in chunk order) do not either.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

class MirtPool {
public:
	static MirtPool& get() {
		static MirtPool p;
		return p;
	}
	unsigned threads() const { return (unsigned)th_.size() + 1u; }

	// fn(chunk) for chunk = 0..nChunks-1, each exactly once, on the pool's threads and the caller's.  Returns when all are
	// done.  Called from inside a region (or while another thread runs one) it runs the chunks on the calling thread.
	void run(uint32_t nChunks, const std::function<void(uint32_t)>& fn) {
		if (nChunks == 0) return;
		if (nChunks == 1 || th_.empty() || inRegion() || !region_.try_lock()) {
			for (uint32_t i = 0; i < nChunks; ++i) fn(i);
			return;
		}
		inRegion() = true;
		auto job = std::make_shared<Job>();
		job->fn = &fn;
		job->n = nChunks;
		{
			std::lock_guard<std::mutex> g(m_);
			job_ = job;
			++gen_;
		}
		cvWork_.notify_all();
		work(*job);
		{
			std::unique_lock<std::mutex> g(m_);
			cvDone_.wait(g, [&]() { return job->active == 0; });
			job_.reset();
		}
		inRegion() = false;
		region_.unlock();
	}

	// fn(i) for i in [0, n), in chunks of `grain` consecutive indices
	template <class F>
	void for_each(uint64_t n, uint32_t grain, F fn) {
		const uint32_t chunks = (uint32_t)((n + grain - 1) / grain);
		run(chunks, [&](uint32_t c) {
			const uint64_t a = (uint64_t)c * grain, b = a + grain < n ? a + grain : n;
			for (uint64_t i = a; i < b; ++i) fn(i);
		});
	}

private:
	struct Job {
		const std::function<void(uint32_t)>* fn = nullptr;
		uint32_t n = 0;
		std::atomic<uint32_t> next{0};
		unsigned active = 0; // workers inside work(), under m_
	};
	std::vector<std::thread> th_;
	std::mutex m_, region_;
	std::condition_variable cvWork_, cvDone_;
	std::shared_ptr<Job> job_;
	uint64_t gen_ = 0;
	bool stop_ = false;

	static bool& inRegion() {
		static thread_local bool v = false;
		return v;
	}
	static void work(Job& j) {
		for (;;) {
			const uint32_t i = j.next.fetch_add(1, std::memory_order_relaxed);
			if (i >= j.n) break;
			(*j.fn)(i);
		}
	}
	MirtPool() {
		unsigned n = std::thread::hardware_concurrency();
		n = n > 16 ? 16 : (n < 1 ? 1 : n);
		if (const char* e = getenv("MIRT_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 64) n = (unsigned)v; } // 1: everything on the calling thread
		for (unsigned i = 1; i < n; ++i) th_.emplace_back([this]() { loop(); });
	}
	~MirtPool() {
		{
			std::lock_guard<std::mutex> g(m_);
			stop_ = true;
		}
		cvWork_.notify_all();
		for (auto& t : th_) t.join();
	}
	void loop() {
		inRegion() = true; // a worker never opens a region of its own
		uint64_t seen = 0;
		for (;;) {
			std::shared_ptr<Job> j;
			{
				std::unique_lock<std::mutex> g(m_);
				cvWork_.wait(g, [&]() { return stop_ || gen_ != seen; });
				if (stop_) return;
				seen = gen_;
				j = job_; // may be gone already: the caller did all the chunks itself
				if (j) ++j->active;
			}
			if (!j) continue;
			work(*j);
			{
				std::lock_guard<std::mutex> g(m_);
				--j->active;
			}
			cvDone_.notify_all();
		}
	}
};
