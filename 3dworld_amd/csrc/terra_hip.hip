// terra_hip.hip -- libterra_hip.so: HIP backend (gfx950 / MI355X) + the C ABI of include/terra.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off terra_hip.hip -o libterra_hip.so
// (see 3dworld_amd/build.py).  There is no host execution path in this library: every entry point needs a HIP device.
#include "terra_kernels.hpp"
#include "terra_fused.hpp"
#include "terra_fz_api.hpp"
#include "terra_simple_paths.hpp"
#include "terra_xfer.hpp"
#include <stdlib.h>

#define TERRA_HIP_CHECK(expr) do {hipError_t const e_ = (expr); if (e_ != hipSuccess) {throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_));}} while (0)

struct hip_backend_t : terra::simple_paths<hip_backend_t> {
	int device = -1;
	hipStream_t stream = nullptr, own_stream = nullptr;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	terra::options_t const *opt = nullptr; // the engine's options (terra_set_option); the members below are copies taken by options_changed()
	bool simple_kernels = false; // "kernels.simple": run the one-thread-per-cell cross-check kernels instead of the LDS-tiled ones
	int num_cus = 256;
	float *tile_pad = nullptr; size_t tile_pad_bytes = 0;
	uint32_t *tile_order = nullptr; size_t tile_order_bytes = 0; // k_tile_erosion's land counts + launch order
	float *vox_p = nullptr; size_t vox_p_bytes = 0;
	bool vox_cols = true; // "voxels.cols": the lane-per-column voxel sine kernel (k_voxel_sines_cols) where it applies (nz a multiple of 4)
	int sg_kc = 27; // "sg.kc": terms per LDS chunk of the heightmap's sine kernel (27: 3 chunks of <= 27 for 8 octaves, 29.7 KB per block; 45: 2 chunks, 48 KB)
	unsigned sg_rowgroup = 4; // "sg.rowgroup": tile rows walked together by k_sine_grid (L2 reuse of table slices)

	static int device_count() {int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n;}
	void init(int dev) {
		device = dev;
		TERRA_HIP_CHECK(hipSetDevice(dev));
		TERRA_HIP_CHECK(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
		stream = own_stream;
		TERRA_HIP_CHECK(hipEventCreate(&ev0)); TERRA_HIP_CHECK(hipEventCreate(&ev1));
		{int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) {num_cus = n;} else {(void)hipGetLastError();}}
		options_changed();
		// LDS-tiled kernels use > 64 KiB of dynamic LDS (160 KiB per CU on gfx950)
		TERRA_HIP_CHECK(hipFuncSetAttribute((void const *)terra::k_tile_erosion, hipFuncAttributeMaxDynamicSharedMemorySize, 96*1024));
		TERRA_HIP_CHECK(hipFuncSetAttribute((void const *)terra::k_tile_shadows_level, hipFuncAttributeMaxDynamicSharedMemorySize, 96*1024));
		TERRA_HIP_CHECK(hipFuncSetAttribute((void const *)terra::k_tile_shadows_flow, hipFuncAttributeMaxDynamicSharedMemorySize, 96*1024));
		TERRA_HIP_CHECK(hipFuncSetAttribute((void const *)terra::k_tile_ao<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64*1024));
		TERRA_HIP_CHECK(hipFuncSetAttribute((void const *)terra::k_tile_ao<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64*1024));
		// the whole context of a tile in one workgroup's LDS (162 408 of the CU's 163 840 bytes); a runtime that refuses the size leaves the band kernel in charge
		ao_tile_ok = hipFuncSetAttribute((void const *)terra::k_tile_ao_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)terra::AOT_LDS) == hipSuccess
			&& hipFuncSetAttribute((void const *)terra::k_tile_ao_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)terra::AOT_LDS) == hipSuccess;
		if (!ao_tile_ok) {(void)hipGetLastError();}
	}
	void options_changed() { // (the engine has drained the stream)
		if (!opt) return;
		simple_kernels = opt->simple_kernels != 0; sg_kc = opt->sg_kc; sg_rowgroup = (unsigned)opt->sg_rowgroup; vox_cols = opt->voxels_cols != 0;
		if (graphs_enabled != (opt->graphs != 0)) {for (graph_slot_t &g : graphs) {graph_drop(g);} graphs_enabled = opt->graphs != 0;}
	}
	~hip_backend_t() {
		if (pin) (void)hipHostFree(pin);
		if (h3_tab) (void)hipFree(h3_tab);
		if (tile_pad) (void)hipFree(tile_pad);
		if (tile_order) (void)hipFree(tile_order);
		if (tile_acc) (void)hipFree(tile_acc);
		if (tile_map) (void)hipFree(tile_map);
		if (vox_p) (void)hipFree(vox_p);
		if (ev0) (void)hipEventDestroy(ev0);
		if (ev1) (void)hipEventDestroy(ev1);
		for (graph_slot_t &g : graphs) {graph_drop(g);}
		if (relay_ev) (void)hipEventDestroy(relay_ev);
		if (side_stream) (void)hipStreamDestroy(side_stream);
		if (own_stream) (void)hipStreamDestroy(own_stream);
	}
	size_t mem_free() {use(); size_t f = 0, t = 0; if (hipMemGetInfo(&f, &t) != hipSuccess) {(void)hipGetLastError(); return ~(size_t)0 >> 1;} return f;}
	void release_scratch() { // the backend's own grow-only buffers (the caller has drained the stream)
		for (void **p : {(void **)&tile_pad, (void **)&tile_order, (void **)&tile_acc, (void **)&tile_map, (void **)&vox_p}) {if (*p) {(void)hipFree(*p); *p = nullptr;}}
		tile_pad_bytes = tile_order_bytes = tile_acc_bytes = vox_p_bytes = 0; tile_map_count = 0;
	}
	void use() {TERRA_HIP_CHECK(hipSetDevice(device));}
	void set_stream(void *s) {sync(); pin_off = 0; stream = s ? (hipStream_t)s : own_stream;} // cached graphs are stream-agnostic (the stream is given at launch)
	void sync() {use(); TERRA_HIP_CHECK(hipStreamSynchronize(stream));}
	void *alloc(size_t bytes) {use(); void *p = nullptr; TERRA_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1)); return p;}
	void free(void *p) {use(); (void)hipFree(p);}
	void fill8(void *p, uint8_t v, size_t count) {use(); if (count) TERRA_HIP_CHECK(hipMemsetAsync(p, v, count, stream));}
	void fill32(void *p, uint32_t v, size_t count) {use(); if (count) TERRA_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)p, (int)v, count, stream));}
	// host buffers are ordinary pageable memory (often stack variables): copies are stream-ordered and then waited for
	void h2d(void *d, void const *h, size_t bytes) {
		use();
		if (bytes >= BIG_XFER) {xfer.submit(device, stream, const_cast<void *>(h), d, bytes, true, [this](hipEvent_t e) {return record_via_side(e);}); xfer.wait_all(); return;} // (the host waits: later kernels of any stream see the data)
		TERRA_HIP_CHECK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream)); TERRA_HIP_CHECK(hipStreamSynchronize(stream));
	}
	// small parameter blocks (tile references, per-column constants, dependency orders): staged through a pinned ring and copied asynchronously, stream-ordered -- the
	// host does not wait (a pageable hipMemcpyAsync + hipStreamSynchronize per upload was ~10 % of a 0.68 ms tile batch).  The ring drains the stream when it wraps.
	uint8_t *pin = nullptr; size_t pin_bytes = 0, pin_off = 0; bool pin_failed = false; // pin_failed: the pinned allocation was refused once -- not retried on every upload
	void h2d_async(void *d, void const *h, size_t bytes) {
		if (bytes == 0) return;
		use();
		if (!pin && !pin_failed) {pin_bytes = (size_t)8 << 20; if (hipHostMalloc((void **)&pin, pin_bytes, hipHostMallocDefault) != hipSuccess) {pin = nullptr; pin_bytes = 0; pin_failed = true; (void)hipGetLastError();}}
		size_t const need = (bytes + 255) & ~(size_t)255;
		if (!pin || need > pin_bytes/2) {h2d(d, h, bytes); return;}
		if (pin_off + need > pin_bytes) {TERRA_HIP_CHECK(hipStreamSynchronize(stream)); pin_off = 0;}
		memcpy(pin + pin_off, h, bytes);
		TERRA_HIP_CHECK(hipMemcpyAsync(d, pin + pin_off, bytes, hipMemcpyHostToDevice, stream));
		pin_off += need;
	}
	void d2h(void *h, void const *d, size_t bytes) {
		use();
		if (bytes >= BIG_XFER) {download_async(d, h, bytes); download_wait(); return;} // whole grids: banded, K streams, pinned staging (terra_xfer.hpp)
		TERRA_HIP_CHECK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, stream)); TERRA_HIP_CHECK(hipStreamSynchronize(stream));
	}
	// ---- big host <-> device transfers (terra_xfer.hpp): stream-ordered behind the work enqueued so far, asynchronous to the host and to the context's later kernels
	static constexpr size_t BIG_XFER = (size_t)16 << 20;
	terra::xfer_engine_t xfer;
	void download_async(void const *d, void *h, size_t bytes) {use(); xfer.submit(device, stream, h, const_cast<void *>(d), bytes, false, [this](hipEvent_t e) {return record_via_side(e);});}
	void download_wait() {xfer.wait_all();}
	static void *host_alloc(size_t bytes) {void *p = nullptr; if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {(void)hipGetLastError(); return nullptr;} return p;}
	static void host_free(void *p) {if (p) (void)hipHostFree(p);}
	// device-to-device copies on this context's stream: inside the device, and from another context's device (several GPUs in one process, terra_multi.hpp)
	void d2d(void *dst, void const *src, size_t bytes) {use(); TERRA_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));}
	void copy_from_peer(void *dst, hip_backend_t &src_be, void const *src, size_t bytes) {
		use();
		if (src_be.device == device) {TERRA_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));}
		else {TERRA_HIP_CHECK(hipMemcpyPeerAsync(dst, device, src, src_be.device, bytes, stream));} // xGMI when peer access is on, staged through the host otherwise
	}
	std::vector<int> peers_mapped; // devices whose memory this device's kernels may address (hipDeviceEnablePeerAccess succeeded)
	void enable_peer(hip_backend_t &other) { // best effort
		if (other.device == device) return;
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, device, other.device) != hipSuccess || !can) return;
		if (hipSetDevice(device) != hipSuccess) return;
		hipError_t const e = hipDeviceEnablePeerAccess(other.device, 0);
		if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {(void)hipGetLastError(); return;}
		if (std::find(peers_mapped.begin(), peers_mapped.end(), other.device) == peers_mapped.end()) {peers_mapped.push_back(other.device);}
	}
	bool can_map(hip_backend_t const &other) const {return other.device == device || std::find(peers_mapped.begin(), peers_mapped.end(), other.device) != peers_mapped.end();}
	// ---- virtual memory management (terra_dgrid: ONE grid whose row strips live on several GPUs, mapped back to back in one address range).
	// A strip is a physical allocation on its device (hipMemCreate) that can leave the process as a POSIX file descriptor; every process (or every device of one
	// process) reserves a range, maps all strips into it in order and enables access for its own device: kernels then address the whole grid through one plain pointer and
	// rows that live on another GPU are reached over xGMI by ordinary loads and stores.
	hipMemAllocationProp vm_prop() const {
		hipMemAllocationProp p; memset(&p, 0, sizeof(p));
		p.type = hipMemAllocationTypePinned; p.requestedHandleTypes = hipMemHandleTypePosixFileDescriptor;
		p.location.type = hipMemLocationTypeDevice; p.location.id = device;
		return p;
	}
	size_t vm_granularity() {use(); hipMemAllocationProp const p = vm_prop(); size_t g = 0; TERRA_HIP_CHECK(hipMemGetAllocationGranularity(&g, &p, hipMemAllocationGranularityRecommended)); return g ? g : ((size_t)2 << 20);}
	void *vm_create(size_t bytes) {use(); hipMemAllocationProp const p = vm_prop(); hipMemGenericAllocationHandle_t h = nullptr; TERRA_HIP_CHECK(hipMemCreate(&h, bytes, &p, 0)); return (void *)h;}
	int vm_export_fd(void *h) {use(); int fd = -1; TERRA_HIP_CHECK(hipMemExportToShareableHandle((void *)&fd, (hipMemGenericAllocationHandle_t)h, hipMemHandleTypePosixFileDescriptor, 0)); return fd;}
	// The runtime that PyTorch 2.10 brings along (HIP 7.0: what a process that imported torch runs on) takes a POINTER to the descriptor; the system runtime of ROCm 7.2 (what a
	// plain C / C++ process such as 3DWorld links) takes the descriptor itself, as CUDA does, and reports the pointer form as "invalid argument" -- found with
	// tools/bench_native_onegrid.c.  The pointer form is tried first (the value form would make an older runtime dereference a small integer).
	void *vm_import_fd(int fd) {
		use();
		hipMemGenericAllocationHandle_t h = nullptr;
		hipError_t e = hipMemImportFromShareableHandle(&h, (void *)&fd, hipMemHandleTypePosixFileDescriptor);
		int ver = 0;
		if (e == hipErrorInvalidValue && hipRuntimeGetVersion(&ver) == hipSuccess && ver >= 70200000) {
			(void)hipGetLastError();
			h = nullptr;
			e = hipMemImportFromShareableHandle(&h, (void *)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor);
		}
		if (e != hipSuccess) {throw std::runtime_error(std::string("hipMemImportFromShareableHandle (descriptor by pointer and by value): ") + hipGetErrorString(e));}
		return (void *)h;
	}
	void *vm_reserve(size_t total, size_t align) {use(); void *p = nullptr; TERRA_HIP_CHECK(hipMemAddressReserve(&p, total, align, nullptr, 0)); return p;}
	void vm_map(void *base, size_t off, void *h, size_t bytes) {use(); TERRA_HIP_CHECK(hipMemMap((uint8_t *)base + off, bytes, 0, (hipMemGenericAllocationHandle_t)h, 0));}
	static void vm_set_access(void *base, size_t total, int const *devices, size_t n) {
		std::vector<hipMemAccessDesc> d(n);
		for (size_t i = 0; i < n; ++i) {memset(&d[i], 0, sizeof(d[i])); d[i].location.type = hipMemLocationTypeDevice; d[i].location.id = devices[i]; d[i].flags = hipMemAccessFlagsProtReadWrite;}
		TERRA_HIP_CHECK(hipMemSetAccess(base, total, d.data(), n));
	}
	static void vm_unmap(void *base, size_t off, size_t bytes) {(void)hipMemUnmap((uint8_t *)base + off, bytes);}
	static void vm_release(void *h) {if (h) (void)hipMemRelease((hipMemGenericAllocationHandle_t)h);}
	static void vm_free(void *base, size_t total) {if (base) (void)hipMemAddressFree(base, total);}
	// stream-level ordering between contexts (terra_event_*): an event recorded on one context's stream, waited for by another's -- the host never blocks
	void *event_create() {use(); hipEvent_t e = nullptr; TERRA_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); return (void *)e;}
	// An event that another THREAD may wait for on the host (terra_event_synchronize) must not be "last recorded" in a stream that later enters graph capture -- the runtime
	// then refuses the wait and poisons the capture (hipErrorStreamCaptureUnsupported; the erosion schedulers capture their rounds on first use) -- so the record goes through
	// a side stream of the context that never captures: [stream: relay event] -> [side stream: wait relay, record e]
	hipStream_t side_stream = nullptr; hipEvent_t relay_ev = nullptr;
	hipError_t record_via_side(hipEvent_t e) {
		hipError_t r = hipSuccess;
		if (!side_stream) {
			if ((r = hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking)) != hipSuccess) return r;
			if ((r = hipEventCreateWithFlags(&relay_ev, hipEventDisableTiming)) != hipSuccess) return r;
		}
		if ((r = hipEventRecord(relay_ev, stream)) != hipSuccess) return r;
		if ((r = hipStreamWaitEvent(side_stream, relay_ev, 0)) != hipSuccess) return r;
		return hipEventRecord(e, side_stream);
	}
	void event_record(void *e) {use(); TERRA_HIP_CHECK(record_via_side((hipEvent_t)e));}
	void event_wait(void *e) {use(); TERRA_HIP_CHECK(hipStreamWaitEvent(stream, (hipEvent_t)e, 0));}
	static void event_destroy(void *e) {if (e) (void)hipEventDestroy((hipEvent_t)e);}
	static void event_synchronize(void *e) {TERRA_HIP_CHECK(hipEventSynchronize((hipEvent_t)e));}
	void timer_start() {use(); TERRA_HIP_CHECK(hipEventRecord(ev0, stream));}
	float timer_stop() {use(); TERRA_HIP_CHECK(hipEventRecord(ev1, stream)); TERRA_HIP_CHECK(hipEventSynchronize(ev1)); float ms = 0; TERRA_HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1)); return ms;}

	// ---- hipGraph cache: a fixed sequence of small dependent launches (one speculative-erosion round) is captured once and replayed with a
	// single hipGraphLaunch.  `key` = every byte the captured closures depend on; the caller re-captures when it changes.
	struct graph_slot_t {std::vector<uint8_t> key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; uint64_t last_use = 0;};
	std::vector<graph_slot_t> graphs; uint64_t graph_clock = 0; bool capturing = false; bool graphs_enabled = true;
	void graph_drop(graph_slot_t &g) {if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); g.exec = nullptr; g.graph = nullptr; g.key.clear();}
	bool graph_replay(void const *key, size_t n) {
		if (!graphs_enabled) return false;
		for (graph_slot_t &g : graphs) {
			if (g.exec && g.key.size() == n && memcmp(g.key.data(), key, n) == 0) {use(); g.last_use = ++graph_clock; TERRA_HIP_CHECK(hipGraphLaunch(g.exec, stream)); return true;}
		}
		return false;
	}
	bool graph_begin() { // false: graphs are off, the caller's launches simply run
		if (!graphs_enabled) return false;
		use();
		TERRA_HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
		capturing = true;
		return true;
	}
	void graph_abort() {if (capturing) {hipGraph_t g = nullptr; (void)hipStreamEndCapture(stream, &g); if (g) (void)hipGraphDestroy(g); capturing = false;}}
	void graph_end(void const *key, size_t n) { // instantiate, remember (16 slots -- a tracer context of the one-grid pipeline holds one per grid in flight --, least recently used goes), launch
		hipGraph_t g = nullptr;
		capturing = false;
		TERRA_HIP_CHECK(hipStreamEndCapture(stream, &g));
		hipGraphExec_t ex = nullptr;
		hipError_t const e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
		if (e != hipSuccess) {(void)hipGraphDestroy(g); throw std::runtime_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(e));}
		graph_slot_t *slot = nullptr;
		if (graphs.size() < 16) {graphs.emplace_back(); slot = &graphs.back();}
		else {slot = &graphs[0]; for (graph_slot_t &c : graphs) {if (c.last_use < slot->last_use) slot = &c;} graph_drop(*slot);}
		slot->key.assign((uint8_t const *)key, (uint8_t const *)key + n); slot->graph = g; slot->exec = ex; slot->last_use = ++graph_clock;
		TERRA_HIP_CHECK(hipGraphLaunch(ex, stream));
	}

	template<class F> void launch(size_t n, F f, int block = 256) {
		if (n == 0) return;
		use();
		size_t const nblocks = (n + block - 1)/block;
		if (nblocks > 0x7FFFFFFFull) throw std::invalid_argument("launch: grid too large");
		hipLaunchKernelGGL(terra::k_generic<F>, dim3((unsigned)nblocks), dim3(block), 0, stream, n, f);
		TERRA_HIP_CHECK(hipGetLastError());
	}

	template<class F> void launch_waves(size_t n, F f) {
		if (n == 0) return;
		use();
		if (n > 0x7FFFFFFFull) throw std::invalid_argument("launch_waves: grid too large");
		hipLaunchKernelGGL(terra::k_waves<F>, dim3((unsigned)n), dim3(64), 0, stream, f, 0u);
		TERRA_HIP_CHECK(hipGetLastError());
	}

	template<class F> void launch_waves_lean(size_t n, F f) { // one 64-lane workgroup per item with the lean droplet scratch (terra_kernels.hpp: k_waves_lean)
		if (n == 0) return;
		use();
		if (n > 0x7FFFFFFFull) throw std::invalid_argument("launch_waves: grid too large");
		hipLaunchKernelGGL(terra::k_waves_lean<F>, dim3((unsigned)n), dim3(64), 0, stream, f);
		TERRA_HIP_CHECK(hipGetLastError());
	}

	template<class F> void launch_waves_nolds(size_t n, F f) { // one 64-lane workgroup per item, no LDS scratch (does not compete with LDS-heavy kernels for a CU)
		if (n == 0) return;
		use();
		if (n > 0x7FFFFFFFull) throw std::invalid_argument("launch_waves: grid too large");
		hipLaunchKernelGGL(terra::k_waves_nolds<F>, dim3((unsigned)n), dim3(64), 0, stream, f);
		TERRA_HIP_CHECK(hipGetLastError());
	}

	// TERRA_GEN_FAST (terra_fused.hpp: k_sine_grid_h3): split the f32 tables of J (k-major, rows padded to nxp / nyp) into scaled half-precision pairs and run the contraction on
	// the half-precision matrix pipe.  amax_y / amax_x: the largest magnitude the row / column table can hold.  Returns false when the tables are too large for its 32-bit offsets.
	void *h3_tab = nullptr; size_t h3_tab_bytes = 0;
	static float h3_scale(float amax) {int e = 0; if (amax > 0.0f && amax < 3.0e38f) {(void)frexpf(amax, &e);} return ldexpf(1.0f, 14 - e);} // amax < 2^e: scaled magnitudes stay below 2^14
	// vox_tab: (voxels) noise_gen_3d's [entry][nsines] sine table with the field's nx, ny -- the split tables are then made straight from it (J.xt / J.yt unused)
	template<int KIND> bool sine_grid_h3(terra::sgf_job_t J, float amax_x, float amax_y, float const *vox_tab = nullptr, uint32_t vnx = 0, uint32_t vny = 0) {
		int const nk = (J.kstart < J.kend) ? J.kend - J.kstart : 0;
		uint32_t const nchunks = (uint32_t)((3*nk + 15)/16);
		if ((uint64_t)std::max(J.nxp, J.nyp)*32u*(nchunks + 1) >= 0xFFFFFFFFull) return false;
		size_t const bx = (size_t)nchunks*J.nxp*32u, by = (size_t)nchunks*J.nyp*32u;
		if (bx + by + 64 > h3_tab_bytes) {if (h3_tab) {sync(); (void)hipFree(h3_tab);} TERRA_HIP_CHECK(hipMalloc(&h3_tab, bx + by + 64)); h3_tab_bytes = bx + by + 64;}
		float const sx = h3_scale(amax_x), sy = h3_scale(amax_y);
		J.xh = h3_tab; J.yh = (char *)h3_tab + bx; J.nchunks = nchunks; J.unscale = 1.0f/(sx*sy);
		if (nchunks && vox_tab) {
			hipLaunchKernelGGL((terra::k_split_voxel_table<false, terra::VOX_SINES>), dim3((J.nxp + 255)/256), dim3(256), 0, stream, vox_tab, vnx, vny, J.nx, J.nxp, sx, (terra::sgh_u4 *)J.xh);
			hipLaunchKernelGGL((terra::k_split_voxel_table<true, terra::VOX_SINES>),  dim3((J.nyp + 255)/256), dim3(256), 0, stream, vox_tab, vnx, vny, J.nx, J.nyp, sy, (terra::sgh_u4 *)J.yh);
		}
		else if (nchunks) {
			hipLaunchKernelGGL(terra::k_split_table<false>, dim3((unsigned)(((size_t)nchunks*J.nxp + 255)/256)), dim3(256), 0, stream, J.xt, J.nxp, J.kstart, J.kend, nchunks, sx, (terra::sgh_u4 *)J.xh);
			hipLaunchKernelGGL(terra::k_split_table<true>,  dim3((unsigned)(((size_t)nchunks*J.nyp + 255)/256)), dim3(256), 0, stream, J.yt, J.nyp, J.kstart, J.kend, nchunks, sy, (terra::sgh_u4 *)J.yh);
		}
		if (KIND == terra::SGF_VOXELS && J.nx <= 64 && (J.nyp % 256u) == 0) {J.narrow = 1; J.ntx = 1; J.nty = J.nyp/256u;} // (rows beyond ny: zero table rows, nothing stored)
		unsigned const nb = J.ntx*J.nty, grid = ((nb + 7)/8)*8;
		hipLaunchKernelGGL(terra::k_sine_grid_h3<KIND>, dim3(grid), dim3(256), 0, stream, J);
		TERRA_HIP_CHECK(hipGetLastError());
		return true;
	}
	// mm (optional): device uint32[2] pre-set to 0xFFFFFFFF receiving min f2ord(z) / min ~f2ord(z); returns false when the caller must run minmax() itself
	bool sine_grid(terra::grid_job_t const &job, terra::noise_consts_t const &nc, terra::sin_lut_t const &L, float const *xt, float const *yt, float const *smx, float const *smy, float *out, uint32_t *mm) {
		if (simple_kernels) {sine_grid_simple(job, nc, L, xt, yt, smx, smy, out); return false;}
		use();
		unsigned const ntx = job.nxp/terra::SG_BX, nty = (job.ny + terra::SG_BY - 1)/terra::SG_BY;
		unsigned const nb = ntx*nty, grid = ((nb + 7)/8)*8;
		if (job.fused) { // TERRA_GEN_FUSED: the sum on the f32 matrix pipe, persistent blocks (two per CU: 225 registers per lane)
			terra::sgf_job_t J; memset(&J, 0, sizeof(J));
			J.xt = xt; J.yt = yt; J.smx = smx; J.smy = smy; J.out = out; J.mm = mm; J.nx = job.nx; J.ny = job.ny; J.nxp = job.nxp; J.nyp = job.nyp; J.ntx = ntx; J.nty = nty; J.rowgroup = sg_rowgroup;
			J.kstart = job.kstart; J.kend = terra::F_TABLE_SIZE; J.glaciate = (job.glaciate && nc.glaciate) ? 1 : 0; J.sine_mag = (job.glaciate && job.use_sine_mag) ? 1 : 0;
			J.zmax_est = nc.zmax_est; J.zmax_est2 = nc.zmax_est2; J.zmax_est2_inv = nc.zmax_est2_inv; J.sine_offset = job.sine_offset;
			if (job.fused == 2 && sine_grid_h3<terra::SGF_GRID>(J, 1.0f, job.fast_amax)) return true; // TERRA_GEN_FAST (|SINF| <= 1 bounds the x table)
			hipLaunchKernelGGL(terra::k_sine_grid_mx<terra::SGF_GRID>, dim3(std::min(grid, (unsigned)(2*num_cus + 7)/8*8)), dim3(256), 0, stream, J);
			TERRA_HIP_CHECK(hipGetLastError());
			return true;
		}
		terra::sg_tiles_t const tl{nullptr, nullptr, 0, sg_rowgroup, 0, 0, 0, 0xFFFFFFFFu, 0, 0xFFFFFFFFu, 0};
		if (job.plain_only) {
			auto const go = [&](auto kern) {hipLaunchKernelGGL(kern, dim3(grid), dim3(terra::SG_THREADS), 0, stream, job, nc, L, xt, yt, smx, smy, out, ntx, nty, mm, tl);};
			if (sg_kc == 27) go(terra::k_sine_grid<false, false, 27>); else if (sg_kc == 20) go(terra::k_sine_grid<false, false, 20>); else go(terra::k_sine_grid<false, false, 45>);
		}
		else                {hipLaunchKernelGGL((terra::k_sine_grid<false, true>),  dim3(grid), dim3(terra::SG_THREADS), 0, stream, job, nc, L, xt, yt, smx, smy, out, ntx, nty, mm, tl);}
		TERRA_HIP_CHECK(hipGetLastError());
		return true;
	}
	bool noise_grid(terra::grid_job_t const &job, terra::noise_consts_t const &nc, terra::sin_lut_t const &L, float const *smx, float const *smy, float *out, uint32_t *mm, uint32_t const *nlut) {
		if (simple_kernels) {noise_grid_simple(job, nc, L, smx, smy, out); return false;}
		use();
		if (job.fused) {TERRA_HIP_CHECK((hipError_t)terra_fz_noise_grid(job.mode, &job, &nc, &L, smx, smy, out, mm, nlut, (void *)stream)); return true;} // TERRA_GEN_FUSED: the contraction-allowed build (terra_fz.hip)
		dim3 const grid((job.nx + 127)/128, (job.ny + terra::NG_ROWS - 1)/terra::NG_ROWS), block(256); // 64 lanes x 2 cells per row segment, 4 rows at a time, NG_ROWS rows per block
		terra::noise_oct_t const oc = terra::make_noise_oct(nc);
		switch (job.mode) {
		case terra::MGEN_PERLIN:      hipLaunchKernelGGL(terra::k_noise_grid<terra::MGEN_PERLIN>,      grid, block, 0, stream, job, nc, L, smx, smy, out, mm, nlut, oc); break;
		case terra::MGEN_DWARP_GPU:   hipLaunchKernelGGL(terra::k_noise_grid<terra::MGEN_DWARP_GPU>,   grid, block, 0, stream, job, nc, L, smx, smy, out, mm, nlut, oc); break;
		case terra::MGEN_SIMPLEX_GPU: hipLaunchKernelGGL(terra::k_noise_grid<terra::MGEN_SIMPLEX_GPU>, grid, block, 0, stream, job, nc, L, smx, smy, out, mm, nlut, oc); break;
		default:                      hipLaunchKernelGGL(terra::k_noise_grid<terra::MGEN_SIMPLEX>,     grid, block, 0, stream, job, nc, L, smx, smy, out, mm, nlut, oc); break;
		}
		TERRA_HIP_CHECK(hipGetLastError());
		return true;
	}
	int32_t *tile_map = nullptr; size_t tile_map_count = 0;
	// may a band of the tiles' fields be evaluated instead of whole squares?  Only where tile_grid takes the packed epilogue of the plain sine kernel (the one that knows bands)
	bool tile_band_ok(uint32_t n, uint32_t nux, uint32_t nuy, uint32_t twx, bool unique_tiles, bool plain_only, int md) const {
		return !simple_kernels && md == terra::MGEN_SINE && unique_tiles && plain_only && (uint64_t)n*2 >= (uint64_t)nux*nuy && ((nux*twx) & 3u) == 0 && (!opt || opt->sg_kc_tiles == 27) && (!opt || opt->ao_bands != 0);
	}
	void tile_grid(uint32_t n, terra::tile_ref_pod_t const *refs, uint32_t nux, uint32_t nuy, float const *xt, float const *yt, uint32_t nxpv, uint32_t nypv, float const *d_sm, float const *d_m0,
		int md, int shp, int kstart, bool use_sm, float so, terra::noise_consts_t const &nc, terra::sin_lut_t const &L, float dxv, float dyv, float *zvals, bool plain_only, uint32_t tw, bool unique_tiles, bool glaciate = true, uint32_t const *nlut = nullptr, int fused = 0, float fast_amax = 0.0f,
		terra::tile_band_t const *band = nullptr) // band (tile_band_ok() said yes): tw = the band's cells per tile in x
	{
		// sine mode and a batch that fills at least half of (distinct tile columns) x (distinct tile rows): ONE LDS-tiled k_sine_grid launch over the
		// virtual grid, scattered into the per-tile layout.  Sparse batches and the fBm modes are per-cell anyway.
		if (!simple_kernels && md != terra::MGEN_SINE) { // fBm modes: per-cell work, two cells per lane
			use();
			terra::grid_job_t job;
			job.mx0 = 0; job.my0 = 0; job.mdx = dxv; job.mdy = dyv; job.nx = job.ny = tw; job.nxp = nxpv; job.nyp = nypv;
			job.mode = md; job.shape = shp; job.kstart = kstart; job.glaciate = glaciate ? 1 : 0; job.use_sine_mag = use_sm ? 1 : 0; job.sine_offset = so; job.plain_only = 0;
			if (fused) {TERRA_HIP_CHECK((hipError_t)terra_fz_noise_tiles(refs, n, nux, d_sm, d_m0, &job, &nc, &L, zvals, tw, nlut, (void *)stream)); return;} // "gen.fused"
			terra::noise_oct_t const oc = terra::make_noise_oct(nc);
			size_t const threads = (size_t)n*tw*((tw + 1)/2);
			dim3 const grid((unsigned)((threads + 255)/256)), block(256);
			switch (md) {
			case terra::MGEN_PERLIN:      hipLaunchKernelGGL(terra::k_noise_tiles<terra::MGEN_PERLIN>,      grid, block, 0, stream, refs, n, nux, d_sm, d_m0, job, nc, L, zvals, tw, nlut, oc); break;
			case terra::MGEN_DWARP_GPU:   hipLaunchKernelGGL(terra::k_noise_tiles<terra::MGEN_DWARP_GPU>,   grid, block, 0, stream, refs, n, nux, d_sm, d_m0, job, nc, L, zvals, tw, nlut, oc); break;
			case terra::MGEN_SIMPLEX_GPU: hipLaunchKernelGGL(terra::k_noise_tiles<terra::MGEN_SIMPLEX_GPU>, grid, block, 0, stream, refs, n, nux, d_sm, d_m0, job, nc, L, zvals, tw, nlut, oc); break;
			default:                      hipLaunchKernelGGL(terra::k_noise_tiles<terra::MGEN_SIMPLEX>,     grid, block, 0, stream, refs, n, nux, d_sm, d_m0, job, nc, L, zvals, tw, nlut, oc); break;
			}
			TERRA_HIP_CHECK(hipGetLastError());
			return;
		}
		if (simple_kernels || md != terra::MGEN_SINE || !unique_tiles || (uint64_t)n*2 < (uint64_t)nux*nuy || ((uintptr_t)zvals & 7)) {tile_grid_simple(n, refs, nux, nuy, xt, yt, nxpv, nypv, d_sm, d_m0, md, shp, kstart, use_sm, so, nc, L, dxv, dyv, zvals, tw, glaciate, fused); return;}
		use();
		size_t const cnt = (size_t)nux*nuy;
		if (cnt > tile_map_count) {if (tile_map) {sync(); (void)hipFree(tile_map);} TERRA_HIP_CHECK(hipMalloc((void **)&tile_map, cnt*sizeof(int32_t))); tile_map_count = cnt;}
		fill32(tile_map, 0xFFFFFFFFu, cnt);
		int32_t *tm = tile_map;
		launch(n, [=] TERRA_LAMBDA (size_t i) {terra::tile_ref_pod_t const r = refs[i]; tm[(size_t)r.yi*nux + r.xi] = (int32_t)i;});
		terra::grid_job_t job;
		job.mx0 = 0; job.my0 = 0; job.mdx = dxv; job.mdy = dyv; job.nx = nux*tw; job.ny = nuy*(band ? band->twy : tw); job.nxp = nxpv; job.nyp = nypv;
		job.mode = terra::MGEN_SINE; job.shape = shp; job.kstart = kstart; job.glaciate = glaciate ? 1 : 0; job.use_sine_mag = use_sm ? 1 : 0; job.sine_offset = so;
		unsigned const ntx = job.nxp/terra::SG_BX, nty = (job.ny + terra::SG_BY - 1)/terra::SG_BY, nb = ntx*nty, grid = ((nb + 7)/8)*8;
		terra::sg_tiles_t tl{tm, d_m0, nux, sg_rowgroup, tw, tw, tw, 0xFFFFFFFFu, 0, 0xFFFFFFFFu, 0};
		if (band) {tl.twy = band->twy; tl.ostride = band->ostride; tl.xsplit = band->xsplit; tl.xgap = band->xgap; tl.ysplit = band->ysplit; tl.ygap = band->ygap; fused = 0;}
		job.plain_only = plain_only ? 1 : 0;
		if (fused && plain_only) { // "gen.fused": the batch's virtual grid on the matrix pipe, scattered into the per-tile layout
			terra::sgf_job_t J; memset(&J, 0, sizeof(J));
			J.xt = xt; J.yt = yt; J.smx = d_sm; J.smy = d_sm + (size_t)nux*tw; J.out = zvals; J.mm = nullptr; J.nx = job.nx; J.ny = job.ny; J.nxp = job.nxp; J.nyp = job.nyp; J.ntx = ntx; J.nty = nty; J.rowgroup = sg_rowgroup;
			J.kstart = kstart; J.kend = terra::F_TABLE_SIZE; J.glaciate = (glaciate && nc.glaciate) ? 1 : 0; J.sine_mag = (glaciate && use_sm) ? 1 : 0;
			J.zmax_est = nc.zmax_est; J.zmax_est2 = nc.zmax_est2; J.zmax_est2_inv = nc.zmax_est2_inv; J.sine_offset = so;
			J.tile_map = tm; J.nux = nux; J.tw = tw;
			if (fused == 2 && sine_grid_h3<terra::SGF_TILES>(J, 1.0f, fast_amax)) return;
			hipLaunchKernelGGL(terra::k_sine_grid_mx<terra::SGF_TILES>, dim3(std::min(grid, (unsigned)(2*num_cus + 7)/8*8)), dim3(256), 0, stream, J);
			TERRA_HIP_CHECK(hipGetLastError());
			return;
		}
		int const kc_tiles = opt ? opt->sg_kc_tiles : 27; // 27: three chunks, 29.7 KB per block (measured on the 64 x 64 batch: 291.8 -> 283.9 us, the 201-wide AO context 735 -> 706 us); "sg.kc_tiles" 45: two chunks, 48 KB.  The same sum either way
		if (plain_only && kc_tiles == 27) {hipLaunchKernelGGL((terra::k_sine_grid<true, false, 27>), dim3(grid), dim3(terra::SG_THREADS), 0, stream, job, nc, L, xt, yt, d_sm, d_sm + (size_t)nux*tw, zvals, ntx, nty, (uint32_t *)nullptr, tl);}
		else if (plain_only) {hipLaunchKernelGGL((terra::k_sine_grid<true, false>), dim3(grid), dim3(terra::SG_THREADS), 0, stream, job, nc, L, xt, yt, d_sm, d_sm + (size_t)nux*tw, zvals, ntx, nty, (uint32_t *)nullptr, tl);}
		else            {hipLaunchKernelGGL((terra::k_sine_grid<true, true>),  dim3(grid), dim3(terra::SG_THREADS), 0, stream, job, nc, L, xt, yt, d_sm, d_sm + (size_t)nux*tw, zvals, ntx, nty, (uint32_t *)nullptr, tl);}
		TERRA_HIP_CHECK(hipGetLastError());
	}
	// the whole batch in one dataflow launch (k_tile_shadows_flow); sync_words: the ticket counter, zeroed here.  false: use the per-level launches.  Leaves SHADOW_EDGE_PUB set in `out`
	// the lane order of the shadow kernels (terra::shadow_lane_order) for the last light / tile geometry seen: the per-level launches of a batch ask for it once per level
	terra::shadow_consts_t lanes_key; uint32_t lanes_np = 0; bool lanes_ok = false; terra::shadow_lanes_t lanes_cached;
	bool shadow_lanes(terra::shadow_consts_t const &c, uint32_t np, terra::shadow_lanes_t &lanes) {
		if (lanes_np != np || memcmp(&lanes_key, &c, sizeof(c)) != 0) {
			memset(&lanes_key, 0, sizeof(lanes_key)); memcpy(&lanes_key, &c, sizeof(c)); lanes_np = np;
			lanes_ok = terra::shadow_lane_order(c, np, terra::SH_LEVEL_THREADS, lanes_cached.path);
		}
		lanes = lanes_cached;
		return lanes_ok;
	}
	bool tile_shadows_flow(terra::shadow_consts_t const &c, uint32_t ntiles, uint32_t nslots, uint32_t const *ord, int32_t const *adj, float const *z, unsigned long long *out, uint8_t *sm, uint32_t np, uint32_t *sync_words) {
		if (simple_kernels || ((uintptr_t)sm & 3) || (opt && opt->shadows_levels)) return false;
		terra::shadow_lanes_t lanes;
		if (!shadow_lanes(c, np, lanes)) return false;
		use();
		fill32(sync_words, 0u, 1); // the ticket (the edge arrays were zeroed by the caller: no stale `published` bit)
		unsigned const grid = std::min<unsigned>(ntiles, (unsigned)(2*num_cus));
		hipLaunchKernelGGL(terra::k_tile_shadows_flow, dim3(grid), dim3(terra::SH_LEVEL_THREADS), terra::SH_LEVEL_LDS, stream, c, nslots, ntiles, ord, adj, z, out, sm, np, sync_words, lanes);
		TERRA_HIP_CHECK(hipGetLastError());
		return true;
	}
	void tile_shadows(terra::shadow_consts_t const &c, uint32_t cnt, uint32_t const *ord, int32_t const *adj, uint32_t n, float const *z, unsigned long long *out, uint8_t *sm, uint32_t np) {
		terra::shadow_lanes_t lanes;
		if (simple_kernels || ((uintptr_t)sm & 3) || !shadow_lanes(c, np, lanes)) {tile_shadows_simple(c, cnt, ord, adj, n, z, out, sm, np); return;} // the block ORs its mask out a word at a time
		use();
		hipLaunchKernelGGL(terra::k_tile_shadows_level, dim3(cnt), dim3(terra::SH_LEVEL_THREADS), terra::SH_LEVEL_LDS, stream, c, n, ord, adj, z, out, sm, np, lanes);
		TERRA_HIP_CHECK(hipGetLastError());
	}
	bool ao_tile_ok = false;
	void tile_ao(uint32_t n, float const *z, float const *ctx, uint8_t *ao, float dz, bool own) {
		if (simple_kernels) {tile_ao_simple(n, z, ctx, ao, dz, own); return;}
		use();
		if (ao_tile_ok && (!opt || opt->ao_whole)) { // one workgroup per tile, the context staged once
			unsigned const grid = std::min<unsigned>(n, (unsigned)num_cus); // persistent: a CU holds one of these workgroups
			if (own) {hipLaunchKernelGGL(terra::k_tile_ao_tile<true>, dim3(grid), dim3(terra::AOT_THREADS), terra::AOT_LDS, stream, z, ctx, ao, dz, n);}
			else {hipLaunchKernelGGL(terra::k_tile_ao_tile<false>, dim3(grid), dim3(terra::AOT_THREADS), terra::AOT_LDS, stream, z, ctx, ao, dz, n);}
			TERRA_HIP_CHECK(hipGetLastError());
			return;
		}
		unsigned const nbands = (terra::AO_TEX + terra::AO_BAND - 1)/terra::AO_BAND;
		size_t const lds = (size_t)(terra::AO_BAND + terra::AO_RL)*terra::AO_CS*sizeof(float);
		if (own) {hipLaunchKernelGGL(terra::k_tile_ao<true>, dim3(n*nbands), dim3(terra::AO_THREADS), lds, stream, z, ctx, ao, dz);}
		else {hipLaunchKernelGGL(terra::k_tile_ao<false>, dim3(n*nbands), dim3(terra::AO_THREADS), lds, stream, z, ctx, ao, dz);}
		TERRA_HIP_CHECK(hipGetLastError());
	}
	uint32_t *tile_acc = nullptr; size_t tile_acc_bytes = 0; // k_tile_post's per-tile accumulators
	void tile_post(uint32_t n, terra::tile_ref_pod_t const *refs, float const *z, terra_tile_stats *st, uint8_t *nm, float *mnz, float wpz, float rad_c, float dxv, float dyv, float dxy) {
		float const c2 = dxy*dxy;
		// k_tile_post takes min_normal_z from the largest |n|^2 and never looks at get_norm's "mag < TOLERANCE" branch: a texel's mag is >= sqrtf(dxdy*dxdy) (a sum that only grows, monotone roundings)
		bool const normalized = !(sqrtf(c2) < 1.0E-12f) && dxy > 0.0f;
		if (simple_kernels || ((uintptr_t)z & 15) || ((uintptr_t)nm & 3) || !normalized) {tile_post_simple(n, refs, z, st, nm, mnz, wpz, rad_c, dxv, dyv, dxy); return;} // the LDS staging reads 16 bytes at a time, texels are stored as words
		use();
		uint32_t flat_word; // the texel of a flat cell (n = (+-0, +-0, dxdy): the ocean floor), by the reference's statements
		{
			float nv[3]; terra::tile_normal_v(0.0f, 0.0f, 0.0f, dxv, dyv, dxy, nv);
			flat_word = (uint32_t)(uint8_t)(127.0*((double)nv[0] + 1.0)) | ((uint32_t)(uint8_t)(127.0*((double)nv[1] + 1.0)) << 8) | ((uint32_t)(uint8_t)(127.0*((double)nv[2] + 1.0)) << 16);
		}
		size_t const bytes = (size_t)n*terra::TP_ACC*sizeof(uint32_t);
		if (bytes > tile_acc_bytes) {if (tile_acc) {sync(); (void)hipFree(tile_acc);} TERRA_HIP_CHECK(hipMalloc((void **)&tile_acc, bytes)); tile_acc_bytes = bytes;}
		hipLaunchKernelGGL(terra::k_tile_post_init, dim3((n*terra::TP_ACC + 255)/256), dim3(256), 0, stream, tile_acc, n);
		hipLaunchKernelGGL(terra::k_tile_post, dim3(n*4), dim3(terra::TP_THREADS), 0, stream, refs, z, st, nm, tile_acc, wpz, dxv, dyv, dxy, c2, flat_word);
		hipLaunchKernelGGL(terra::k_tile_post_final, dim3((n + 255)/256), dim3(256), 0, stream, refs, n, st, mnz, tile_acc, rad_c, dxy, nm ? 1 : 0);
		TERRA_HIP_CHECK(hipGetLastError());
	}
	void tile_erosion(uint32_t n, float *zvals, terra::erosion_consts_t const &ec, uint32_t iters) {
		use();
		size_t const lds = (size_t)ec.NX*ec.NY*sizeof(float);
		// default: the whole clamp-padded tile resident in LDS (76 KB, 2 tiles per CU; measured 126 ms for 4096 tiles x 1000 droplets);
		// option "tile_erosion" = "window": a 32x32 LDS window over an HBM/L2-resident copy (10 KB, ~15 tiles per CU; 141 ms: the batch is bound by its
		// heaviest land tiles, ~50k dependent droplet steps each, not by occupancy).  Grids too large for LDS always use the window.
		bool const use_lds = !(opt && opt->tile_erosion_window) && lds <= 96*1024;
		if (!use_lds) {
			size_t const bytes = (size_t)n*lds;
			if (bytes > tile_pad_bytes) {if (tile_pad) {sync(); (void)hipFree(tile_pad);} TERRA_HIP_CHECK(hipMalloc((void **)&tile_pad, bytes)); tile_pad_bytes = bytes;}
			if (simple_kernels) {tile_erosion_simple(n, zvals, ec, iters, tile_pad);} // cross-check path: one scalar lane per tile
			else {tile_erosion_windowed(n, zvals, ec, iters, tile_pad);}
			return;
		}
		uint32_t const *d_order = nullptr, *d_landc = nullptr;
		if (n > 512 && (uint64_t)n*iters >= (1u << 16)) { // more tiles than the chip holds at once (2 per CU): longest predicted chains first
			size_t const bytes = (size_t)n*2*sizeof(uint32_t);
			if (bytes > tile_order_bytes) {if (tile_order) {sync(); (void)hipFree(tile_order);} TERRA_HIP_CHECK(hipMalloc((void **)&tile_order, bytes)); tile_order_bytes = bytes;}
			uint32_t *d_land = tile_order, *d_ord = tile_order + n;
			fill32(d_land, 0, n);
			hipLaunchKernelGGL(terra::k_tile_land_cells, dim3(n), dim3(256), 0, stream, zvals, (uint32_t)(ec.xsize*ec.ysize), ec.water_thresh, d_land);
			TERRA_HIP_CHECK(hipGetLastError());
			if (n <= 65536) { // the order is made on the device (rank by counting: n^2 / 2^12 block-steps, 16 M comparisons for the 64 x 64 batch): the call never waits for the stream
				hipLaunchKernelGGL(terra::k_tile_order_by_land, dim3((n + 255)/256), dim3(256), 0, stream, d_land, n, d_ord);
				TERRA_HIP_CHECK(hipGetLastError());
			}
			else {
				std::vector<uint32_t> land(n), ord(n);
				d2h(land.data(), d_land, (size_t)n*4);
				for (uint32_t i = 0; i < n; ++i) {ord[i] = i;}
				std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {return land[a] > land[b];});
				h2d(d_ord, ord.data(), (size_t)n*4);
			}
			d_order = d_ord; d_landc = d_land;
		}
		hipLaunchKernelGGL(terra::k_tile_erosion, dim3(n), dim3(64), lds, stream, zvals, ec, iters, d_order, d_landc);
		TERRA_HIP_CHECK(hipGetLastError());
	}
	void minmax(float const *vals, size_t n, uint32_t *d) {
		if (simple_kernels || ((uintptr_t)vals & 15)) {minmax_simple(vals, n, d); return;}
		use();
		unsigned const blocks = (unsigned)std::min<size_t>((n/4 + 255)/256 + 1, 256*8);
		hipLaunchKernelGGL(terra::k_minmax, dim3(blocks), dim3(256), 0, stream, vals, n, d);
		TERRA_HIP_CHECK(hipGetLastError());
	}
	void quantize16(float const *vals, size_t n, float val_add, float val_div, uint8_t *pix) {
		size_t const n8 = ((simple_kernels || ((uintptr_t)vals & 15) || ((uintptr_t)pix & 15)) ? 0 : n/8);
		if (n8) {
			use();
			if ((n8 + 255)/256 > 0x7FFFFFFFull) throw std::invalid_argument("quantize16: grid too large");
			hipLaunchKernelGGL(terra::k_quantize16, dim3((unsigned)((n8 + 255)/256)), dim3(256), 0, stream, vals, n8, val_add, val_div, (terra::st_u4 *)pix);
			TERRA_HIP_CHECK(hipGetLastError());
		}
		quantize16_simple(vals + n8*8, n - n8*8, val_add, val_div, pix + n8*16); // the tail (or everything, unaligned / cross-check)
	}
	bool tile_weights(terra::landscape_consts_t const &c, terra::tile_ref_pod_t const *refs, uint32_t n, float const *zvals, float const *noise, float const *params, uint32_t *w32, terra::grass_block_pod_t *blocks, uint8_t *any_grass) {
		if (simple_kernels || (uint64_t)n*terra::WK_BANDS > 0x7FFFFFFFull) return false;
		use();
		hipLaunchKernelGGL(terra::k_tile_weights, dim3(n*terra::WK_BANDS), dim3(256), 0, stream, c, refs, zvals, noise, params, w32, blocks, any_grass);
		TERRA_HIP_CHECK(hipGetLastError());
		return true;
	}
	void voxel_noise(float *out, size_t nvox, terra::vox_noise_job_t const &J, bool perlin, bool fused, uint32_t const *lut3) {
		if (simple_kernels) {voxel_noise_simple(out, nvox, J, perlin); return;}
		if (nvox == 0) return;
		use();
		// "gen.fused" has no kernel here: glm's 3-D lattice noise picks its gradients by the SIGN of expressions that are exactly zero at some of the hash's 49 / 289 values
		// (h = 1 - |x| - |y| in simplex(vec3), gz = 0.5 - |gx| - |gy| in perlin(vec3); noise.inl:149-157,690-700), so one contracted rounding flips a gradient: measured 0.146
		// on a 40 x 24 x 64 Perlin field for 1.2x.  The exact kernel answers (its gradients come from the table the exact code builds).
		(void)fused;
		uint32_t const nzp = (J.nz + 1)/2; // pairs (z, z + 1) per column
		size_t const npairs = (nvox / J.nz)*nzp, per_block = (size_t)256*terra::VN_CHUNKS;
		if ((npairs + per_block - 1)/per_block > 0x7FFFFFFFull) throw std::invalid_argument("voxel_fill: grid too large");
		dim3 const grid((unsigned)((npairs + per_block - 1)/per_block)), block(256);
		if (perlin) {hipLaunchKernelGGL(terra::k_voxel_noise<true>, grid, block, 0, stream, out, npairs, nzp, J, lut3);}
		else        {hipLaunchKernelGGL(terra::k_voxel_noise<false>, grid, block, 0, stream, out, npairs, nzp, J, lut3);}
		TERRA_HIP_CHECK(hipGetLastError());
	}
	void voxel_sines(float *out, uint32_t nx, uint32_t ny, uint32_t nz, float const *d_tab, float zscale, int normalize, int fused = 0, float fast_amax = 0.0f, float const *d_zt = nullptr, uint32_t nzp = 0) {
		if (simple_kernels) {voxel_sines_simple(out, nx, ny, nz, d_tab, zscale, normalize, fused); return;}
		use();
		if (fused && (uint64_t)nx*ny <= 0x7FFFFF00ull) { // "gen.fused": the field as a (columns x 60) x (60 x nz) product on the f32 matrix pipe (terra_fused.hpp)
			size_t const ncol = (size_t)nx*ny;
			uint32_t const nyp = (uint32_t)((ncol + 255)/256*256), nxp = (nz + 127)/128*128; // (256: the narrow form of k_sine_grid_h3 walks 256-row tiles)
			terra::sgf_job_t J; memset(&J, 0, sizeof(J));
			J.out = out; J.nx = nz; J.ny = (uint32_t)ncol; J.nxp = nxp; J.nyp = nyp; J.ntx = nxp/128; J.nty = nyp/128; J.rowgroup = sg_rowgroup;
			J.kstart = 0; J.kend = terra::VOX_SINES; J.zscale = zscale; J.normalize = normalize;
			if (fused == 2 && sine_grid_h3<terra::SGF_VOXELS>(J, 1.0f, fast_amax, d_tab, nx, ny)) return; // TERRA_GEN_FAST (|zv| <= 1; |xv*yv| <= the largest magnitude)
			size_t const np = (size_t)terra::VOX_SINES*((size_t)nyp + nxp);
			if (np*4 > vox_p_bytes) {if (vox_p) {sync(); (void)hipFree(vox_p);} TERRA_HIP_CHECK(hipMalloc((void **)&vox_p, np*4)); vox_p_bytes = np*4;}
			float *const pt = vox_p, *const zt = vox_p + (size_t)terra::VOX_SINES*nyp;
			// PT[k][column] = xv[x][k]*yv[y][k] (the product rounds as in the reference, src/upsurface.cpp:66; only the multiply-add with zv is fused), ZT[k][z]: both k-major, zero padded
			launch(np, [=] TERRA_LAMBDA (size_t i) {
				if (i < (size_t)terra::VOX_SINES*nyp) {
					uint32_t const k = (uint32_t)(i / nyp); size_t const c = i % nyp;
					float v = 0.0f;
					if (c < ncol) {uint32_t const x = (uint32_t)(c % nx), y = (uint32_t)(c / nx); v = __fmul_rn(d_tab[(size_t)x*terra::VOX_SINES + k], d_tab[((size_t)nx + y)*terra::VOX_SINES + k]);}
					pt[i] = v;
				}
				else {
					size_t const j = i - (size_t)terra::VOX_SINES*nyp;
					uint32_t const k = (uint32_t)(j / nxp), z = (uint32_t)(j % nxp);
					zt[j] = (z < nz) ? d_tab[((size_t)nx + ny + z)*terra::VOX_SINES + k] : 0.0f;
				}
			});
			J.xt = zt; J.yt = pt;
			unsigned const nb = J.ntx*J.nty, grid = ((nb + 7)/8)*8;
			hipLaunchKernelGGL(terra::k_sine_grid_mx<terra::SGF_VOXELS>, dim3(std::min(grid, (unsigned)(2*num_cus + 7)/8*8)), dim3(256), 0, stream, J);
			TERRA_HIP_CHECK(hipGetLastError());
			return;
		}
		if ((nz & 3u) == 0 && vox_cols && d_zt) { // a lane per column, the products in registers, the z table as scalar operands (k_voxel_sines_cols): no P array
			size_t const ncolc = (size_t)nx*ny;
			hipLaunchKernelGGL(terra::k_voxel_sines_cols, dim3((unsigned)((ncolc + terra::VC_COLS - 1)/terra::VC_COLS), (nz + terra::VC_Z - 1)/terra::VC_Z), dim3(256), 0, stream, out, nx, ny, nz, nzp, d_tab, d_zt, zscale, normalize);
			TERRA_HIP_CHECK(hipGetLastError());
			return;
		}
		size_t const ncol2 = ((size_t)nx*ny + 1) & ~(size_t)1, np = (ncol2/2)*terra::VX_PSTRIDE + (size_t)nz*terra::VOX_SINES, nprod = (ncol2 + nz)*terra::VOX_SINES; // column pairs + transposed z table
		if (np*4 > vox_p_bytes) {if (vox_p) {sync(); (void)hipFree(vox_p);} TERRA_HIP_CHECK(hipMalloc((void **)&vox_p, np*4)); vox_p_bytes = np*4;}
		hipLaunchKernelGGL(terra::k_voxel_P, dim3((unsigned)((nprod + 255)/256)), dim3(256), 0, stream, vox_p, vox_p + (ncol2/2)*terra::VX_PSTRIDE, nx, ny, nz, d_tab);
		unsigned const block = (nz >= 256) ? 256 : ((nz + 63)/64)*64;
		size_t const ncol = (size_t)nx*ny;
		hipLaunchKernelGGL(terra::k_voxel_sines, dim3((unsigned)((ncol + terra::VX_PER_BLOCK - 1)/terra::VX_PER_BLOCK), (nz + block - 1)/block), dim3(block), 0, stream, out, nx, ny, nz, vox_p + (ncol2/2)*terra::VX_PSTRIDE, vox_p, zscale, normalize);
		TERRA_HIP_CHECK(hipGetLastError());
	}
};
typedef hip_backend_t terra_backend_t;
#include "terra_api_impl.hpp"
