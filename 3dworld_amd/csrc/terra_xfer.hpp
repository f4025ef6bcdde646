// terra_xfer.hpp -- host <-> device transfers of whole grids for the host-pointer entry points (HIP backend only; included by terra_hip.hip).
//
// The reference's callers hand over and receive HOST arrays: mesh_xy_grid_cache_t::build_arrays fills cached_vals (src/mesh_gen.cpp:597-603), apply_erosion works on
// the caller's float* (src/erosion.cpp:14), heightmap_t::proc_gen returns a vector (src/heightmap.cpp:130-151), the map exporter writes pixels (src/map_view.cpp:409-442).
// At 16384^2 that is 1 GiB per direction, and a plain hipMemcpy to pageable memory is one blocking call that the runtime stages through its own small pinned buffer.
// Here a transfer is cut into bands that K worker threads move concurrently, each on its own stream with two pinned staging slots: while band i of a thread is copied
// between its slot and the caller's (pageable) array by the CPU, band i + K is on the PCIe link.  A caller array that is itself pinned (terra_host_alloc, or memory the
// caller registered) is the DMA target directly -- no CPU copy.  The first band waits (on the device) for an event recorded on the context's compute stream, so a download can
// be started right behind the kernels that produce the grid and runs beside whatever the context enqueues next: terra_download_async / terra_download_wait.
#pragma once
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <string>

namespace terra {

struct xfer_engine_t {
	static constexpr size_t BAND = (size_t)8 << 20; // bytes per band
	static constexpr int K = 4;                    // worker threads = streams = pairs of pinned slots (64 MiB of pinned memory per context that ever transfers a big array)
	struct job_t {uint8_t *host; uint8_t *dev; size_t bytes; bool to_device; bool host_pinned; hipEvent_t ready;};
	int device = -1;
	std::mutex mtx; std::condition_variable cv_work, cv_done;
	std::vector<std::thread> threads;
	std::vector<job_t> queue;   // jobs are appended; every worker walks the whole queue in order
	size_t next_job[K] = {0, 0, 0, 0}; // per worker: index of the next job it has not done yet
	size_t finished = 0;         // jobs that ALL workers have completed
	size_t done_by[K] = {0, 0, 0, 0};
	bool quit = false;
	std::string error;

	~xfer_engine_t() {
		{std::lock_guard<std::mutex> l(mtx); quit = true;}
		cv_work.notify_all();
		for (std::thread &t : threads) {if (t.joinable()) t.join();}
		for (job_t &j : queue) {if (j.ready) (void)hipEventDestroy(j.ready);}
	}
	static bool is_pinned(void const *p) {
		hipPointerAttribute_t a; memset(&a, 0, sizeof(a));
		if (hipPointerGetAttributes(&a, p) != hipSuccess) {(void)hipGetLastError(); return false;}
		return a.type == hipMemoryTypeHost;
	}
	void start(int dev) {
		if (!threads.empty()) return;
		device = dev;
		for (int k = 0; k < K; ++k) {threads.emplace_back([this, k]() {worker(k);});}
	}
	void fail(char const *what, hipError_t e) {std::lock_guard<std::mutex> l(mtx); if (error.empty()) {error = std::string(what) + ": " + hipGetErrorString(e);}}
	void worker(int k) {
		hipStream_t st = nullptr; uint8_t *slot[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr};
		bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
		for (int s = 0; s < 2 && ok; ++s) {ok = hipHostMalloc((void **)&slot[s], BAND, hipHostMallocDefault) == hipSuccess && hipEventCreateWithFlags(&ev[s], hipEventDisableTiming) == hipSuccess;}
		if (!ok) {fail("transfer worker set-up", hipGetLastError());}
		for (;;) {
			job_t j;
			{
				std::unique_lock<std::mutex> l(mtx);
				cv_work.wait(l, [&]() {return quit || next_job[k] < queue.size();});
				if (quit) break;
				j = queue[next_job[k]];
			}
			if (ok) {run(j, k, st, slot, ev);}
			{
				std::lock_guard<std::mutex> l(mtx);
				++next_job[k]; ++done_by[k];
				size_t f = done_by[0]; for (int q = 1; q < K; ++q) {f = (done_by[q] < f) ? done_by[q] : f;}
				finished = f;
			}
			cv_done.notify_all();
		}
		for (int s = 0; s < 2; ++s) {if (slot[s]) (void)hipHostFree(slot[s]); if (ev[s]) (void)hipEventDestroy(ev[s]);}
		if (st) (void)hipStreamDestroy(st);
	}
	// worker k moves bands k, k + K, k + 2K, ... of the job
	void run(job_t const &j, int k, hipStream_t st, uint8_t *const slot[2], hipEvent_t const ev[2]) {
		hipError_t e = hipStreamWaitEvent(st, j.ready, 0);
		if (e != hipSuccess) {fail("hipStreamWaitEvent", e); return;}
		size_t const nb = (j.bytes + BAND - 1)/BAND;
		if (j.host_pinned) {
			// the caller's array is the DMA target / source: ONE copy on ONE stream.  (Bands on all K streams -- what the staged path below needs to overlap its CPU copies --
			// only make the DMA queues compete when there is nothing to overlap: the driver's box delivered 42 GB/s that way against 56 for a single hipMemcpy, BENCH_r05
			// detail.end_to_end, and the staged pageable path 49.)
			if (k != 0) return;
			e = j.to_device ? hipMemcpyAsync(j.dev, j.host, j.bytes, hipMemcpyHostToDevice, st) : hipMemcpyAsync(j.host, j.dev, j.bytes, hipMemcpyDeviceToHost, st);
			if (e != hipSuccess) {fail("hipMemcpyAsync", e); return;}
			e = hipStreamSynchronize(st);
			if (e != hipSuccess) {fail("hipStreamSynchronize", e);}
			return;
		}
		if (j.to_device) { // pageable -> slot (CPU) -> device (DMA); the slot is reused once its DMA has completed
			int s = 0; bool used[2] = {false, false};
			for (size_t b = (size_t)k; b < nb; b += K, s ^= 1) {
				size_t const off = b*BAND, n = (j.bytes - off < BAND) ? j.bytes - off : BAND;
				if (used[s]) {e = hipEventSynchronize(ev[s]); if (e != hipSuccess) {fail("hipEventSynchronize", e); return;}}
				memcpy(slot[s], j.host + off, n);
				e = hipMemcpyAsync(j.dev + off, slot[s], n, hipMemcpyHostToDevice, st);
				if (e == hipSuccess) {e = hipEventRecord(ev[s], st);}
				if (e != hipSuccess) {fail("hipMemcpyAsync", e); return;}
				used[s] = true;
			}
			e = hipStreamSynchronize(st);
			if (e != hipSuccess) {fail("hipStreamSynchronize", e);}
			return;
		}
		// device -> slot (DMA) -> pageable (CPU): the DMA of the next band is in flight while this one is copied out
		size_t pend_off[2] = {0, 0}, pend_n[2] = {0, 0}; bool pend[2] = {false, false};
		int s = 0;
		for (size_t b = (size_t)k; b < nb; b += K, s ^= 1) {
			size_t const off = b*BAND, n = (j.bytes - off < BAND) ? j.bytes - off : BAND;
			if (pend[s]) {e = hipEventSynchronize(ev[s]); if (e != hipSuccess) {fail("hipEventSynchronize", e); return;} memcpy(j.host + pend_off[s], slot[s], pend_n[s]); pend[s] = false;}
			e = hipMemcpyAsync(slot[s], j.dev + off, n, hipMemcpyDeviceToHost, st);
			if (e == hipSuccess) {e = hipEventRecord(ev[s], st);}
			if (e != hipSuccess) {fail("hipMemcpyAsync", e); return;}
			pend[s] = true; pend_off[s] = off; pend_n[s] = n;
		}
		for (int q = 0; q < 2; ++q, s ^= 1) { // the two slots still in flight, older first
			if (pend[s]) {e = hipEventSynchronize(ev[s]); if (e != hipSuccess) {fail("hipEventSynchronize", e); return;} memcpy(j.host + pend_off[s], slot[s], pend_n[s]); pend[s] = false;}
		}
	}
	// enqueue; `compute` is the stream whose work so far the transfer must wait for (and, for an upload, the stream that must wait for wait_all() before using the data: the caller does)
	// `record_ready(e)`: records e stream-ordered behind `compute`'s work so far WITHOUT making `compute` the event's last-recording stream -- the workers wait for it later
	// (hipStreamWaitEvent), and by then `compute` may be capturing a graph (the erosion schedulers capture their rounds on first use of a shape): HIP refuses a wait on an
	// event last recorded in a capturing stream and invalidates the capture.  The backend relays through its never-capturing side stream, as terra_event_record does.
	template<class RECORD> void submit(int dev, hipStream_t compute, void *host, void *devp, size_t bytes, bool to_device, RECORD record_ready) {
		if (bytes == 0) return;
		(void)compute;
		start(dev);
		job_t j; j.host = (uint8_t *)host; j.dev = (uint8_t *)devp; j.bytes = bytes; j.to_device = to_device; j.host_pinned = is_pinned(host); j.ready = nullptr;
		if (hipEventCreateWithFlags(&j.ready, hipEventDisableTiming) != hipSuccess || record_ready(j.ready) != hipSuccess) {
			if (j.ready) (void)hipEventDestroy(j.ready);
			throw std::runtime_error("terra transfer: event set-up failed");
		}
		{std::lock_guard<std::mutex> l(mtx); queue.push_back(j);}
		cv_work.notify_all();
	}
	// every transfer submitted so far has landed; throws the first worker error
	void wait_all() {
		std::unique_lock<std::mutex> l(mtx);
		cv_done.wait(l, [&]() {return finished >= queue.size();});
		for (job_t &j : queue) {if (j.ready) (void)hipEventDestroy(j.ready);}
		queue.clear(); finished = 0;
		for (int k = 0; k < K; ++k) {next_job[k] = 0; done_by[k] = 0;}
		if (!error.empty()) {std::string const e = error; error.clear(); throw std::runtime_error("terra transfer: " + e);}
	}
	bool idle() {std::lock_guard<std::mutex> l(mtx); return queue.empty();}
};

} // namespace terra
