// compile-only check of include/terra_cxx.hpp: the mirror keeps the reference's signatures (src/mesh.h:40-44, src/function_registry.h:354)
#include "../include/terra_cxx.hpp"
using namespace terra_cxx;
int main() {
	bool (mesh_xy_grid_cache_t::*ba)(float, float, float, float, unsigned, unsigned, bool, bool, bool) = &mesh_xy_grid_cache_t::build_arrays;
	void (mesh_xy_grid_cache_t::*eg)() = &mesh_xy_grid_cache_t::enable_glaciate;
	float (mesh_xy_grid_cache_t::*ei)(unsigned, unsigned, int, bool) const = &mesh_xy_grid_cache_t::eval_index;
	void (*ae)(float *, int, int, float, unsigned) = &apply_erosion;
	return (ba && eg && ei && ae) ? 0 : 1;
}
