// terra_dgrid.hpp -- ONE grid whose row strips live on several GPUs (include/terra.h: terra_dgrid_*, terra_multi_dgrid_create).
//
// The reference erodes a whole heightmap as one shared array in serial droplet order (apply_erosion, src/erosion.cpp:66-155): its rows cannot be given to different
// GPUs the way heightmap_t::proc_gen's evaluation loop can (src/heightmap.cpp:139-143).  What can be spread is the MEMORY: every rank (or every context of one process)
// owns one strip of rows as a physical allocation on its device, all strips are mapped back to back into one virtual address range on every device (HIP virtual memory
// management; between processes a strip travels as a POSIX file descriptor), and each rank fills its own strip with the noise kernel at HBM speed.  The erosion then runs
// on ONE device over the plain pointer: droplets that start in a remote strip pull their 32 x 32 windows over xGMI and write them back the same way -- the kernels, the
// serial order and therefore every bit of the result are those of the single-GPU call.  Included by terra_api_impl.hpp (both libraries build it; the emulator's
// stand-in is memfd + mmap, so two emulator processes share a grid exactly like two ranks do).
#pragma once

struct terra_dgrid {
	terra_ctx *ctx = nullptr;                // the context (device) this view of the grid belongs to
	std::vector<void *> handles;             // one physical allocation per strip (owned: created here or imported)
	std::vector<size_t> bytes;               // strip sizes (multiples of the granularity)
	std::vector<int> devices;                // devices that get access to the mapping (in-process form: every context's device)
	void *base = nullptr; size_t total = 0; bool mapped = false; // mapped: every strip is mapped AND accessible (set last)
	size_t n_mapped = 0;                     // strips 0 .. n_mapped-1 are mapped into `base` (what dgrid_release / a failed terra_dgrid_map take apart)
	uint32_t local = 0;
};

namespace terra {
inline void dgrid_release(terra_dgrid *g) {
	if (!g) return;
	if (g->base) {
		{size_t off = 0; for (size_t i = 0; i < g->n_mapped; ++i) {terra_backend_t::vm_unmap(g->base, off, g->bytes[i]); off += g->bytes[i];}}
		terra_backend_t::vm_free(g->base, g->total);
	}
	for (void *h : g->handles) {terra_backend_t::vm_release(h);}
	delete g;
}
inline size_t dgrid_total(std::vector<size_t> const &b) {size_t t = 0; for (size_t v : b) t += v; return t;}
} // namespace terra

extern "C" {

size_t terra_dgrid_granularity(terra_ctx *ctx) {
	if (!ctx) return 0;
	try {return ctx->eng.be.vm_granularity();} catch (std::exception const &e) {terra::fail(TERRA_ERR_HIP, e.what()); return 0;}
}
// this rank's view: n_strips strips of strip_bytes[i] each; only strip `local_strip` is allocated here (on ctx's device)
int terra_dgrid_create(terra_ctx *ctx, uint32_t n_strips, const size_t *strip_bytes, uint32_t local_strip, terra_dgrid **out) {
	TERRA_CHECK_CTX
	if (!out) return terra::fail(TERRA_ERR_ARG, "null out");
	*out = nullptr;
	if (!strip_bytes || n_strips == 0 || n_strips > 4096 || local_strip >= n_strips) return terra::fail(TERRA_ERR_ARG, "terra_dgrid_create: bad strip list");
	TERRA_TRY
		size_t const gran = ctx->eng.be.vm_granularity();
		for (uint32_t i = 0; i < n_strips; ++i) {if (strip_bytes[i] == 0 || strip_bytes[i] % gran) throw std::invalid_argument("terra_dgrid_create: every strip must be a non-empty multiple of terra_dgrid_granularity()");}
		terra_dgrid *g = new terra_dgrid();
		g->ctx = ctx; g->local = local_strip; g->bytes.assign(strip_bytes, strip_bytes + n_strips); g->handles.assign(n_strips, nullptr); g->total = terra::dgrid_total(g->bytes);
		g->devices.push_back(ctx->eng.be.device);
		try {g->handles[local_strip] = ctx->eng.be.vm_create(strip_bytes[local_strip]);} catch (...) {terra::dgrid_release(g); throw;}
		*out = g;
	TERRA_CATCH
}
int terra_dgrid_export_fd(terra_dgrid *g, int *fd) {
	if (!g || !fd) return terra::fail(TERRA_ERR_ARG, "null argument");
	TERRA_TRY *fd = g->ctx->eng.be.vm_export_fd(g->handles[g->local]); TERRA_CATCH
}
int terra_dgrid_import_fd(terra_dgrid *g, uint32_t strip, int fd) {
	if (!g || strip >= g->handles.size() || fd < 0) return terra::fail(TERRA_ERR_ARG, "terra_dgrid_import_fd: bad strip or descriptor");
	if (g->handles[strip] || g->mapped) return terra::fail(TERRA_ERR_STATE, "terra_dgrid_import_fd: the strip is already there");
	TERRA_TRY g->handles[strip] = g->ctx->eng.be.vm_import_fd(fd); TERRA_CATCH
}
int terra_dgrid_map(terra_dgrid *g, void **d_base) {
	if (!g || !d_base) return terra::fail(TERRA_ERR_ARG, "null argument");
	if (g->mapped) {*d_base = g->base; return TERRA_OK;}
	for (void *h : g->handles) {if (!h) return terra::fail(TERRA_ERR_STATE, "terra_dgrid_map: a strip has not been imported yet");}
	TERRA_TRY
		auto &be = g->ctx->eng.be;
		size_t const gran = be.vm_granularity();
		if (!g->base) {g->base = be.vm_reserve(g->total, gran);}
		try {
			size_t off = 0;
			for (size_t i = 0; i < g->handles.size(); ++i) {be.vm_map(g->base, off, g->handles[i], g->bytes[i]); g->n_mapped = i + 1; off += g->bytes[i];}
			terra_backend_t::vm_set_access(g->base, g->total, g->devices.data(), g->devices.size());
		} catch (...) { // leave the handle as it was before the call: nothing mapped, the call may be repeated
			size_t off = 0;
			for (size_t i = 0; i < g->n_mapped; ++i) {terra_backend_t::vm_unmap(g->base, off, g->bytes[i]); off += g->bytes[i];}
			g->n_mapped = 0;
			throw;
		}
		g->mapped = true;
		*d_base = g->base;
	TERRA_CATCH
}
void terra_dgrid_destroy(terra_dgrid *g) {if (g) {try {g->ctx->eng.be.sync();} catch (...) {} terra::dgrid_release(g);}}

// the in-process form: strip i on context i's device, one mapping that every context's device may access
int terra_multi_dgrid_create(terra_multi *m, const size_t *strip_bytes, terra_dgrid **out, void **d_base) {
	if (!m || m->ctxs.empty()) return terra::fail(TERRA_ERR_ARG, "null terra_multi");
	if (!strip_bytes || !out || !d_base) return terra::fail(TERRA_ERR_ARG, "null argument");
	*out = nullptr;
	TERRA_TRY
		uint32_t const n = (uint32_t)m->ctxs.size();
		terra_dgrid *g = new terra_dgrid();
		g->ctx = m->ctxs[0]; g->bytes.assign(strip_bytes, strip_bytes + n); g->handles.assign(n, nullptr); g->total = terra::dgrid_total(g->bytes);
		try {
			for (uint32_t i = 0; i < n; ++i) {
				auto &be = m->ctxs[i]->eng.be;
				if (strip_bytes[i] == 0 || strip_bytes[i] % be.vm_granularity()) throw std::invalid_argument("terra_multi_dgrid_create: every strip must be a non-empty multiple of terra_dgrid_granularity()");
				g->handles[i] = be.vm_create(strip_bytes[i]);
				if (std::find(g->devices.begin(), g->devices.end(), be.device) == g->devices.end()) {g->devices.push_back(be.device);}
			}
			int const rc = terra_dgrid_map(g, d_base);
			if (rc != TERRA_OK) {std::string const msg = terra_last_error(); throw std::runtime_error(msg);}
		} catch (...) {terra::dgrid_release(g); throw;}
		*out = g;
	TERRA_CATCH
}

} // extern "C"
