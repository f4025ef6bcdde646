// step-level dataflow bound of the droplet dependency structure (cells: cell<<1|write in access order per droplet)
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
// mode: shift = 0 cell level, 3 = 8x8 block level.  whole != 0: a droplet starts when every lower writer of anything it touches has FINISHED (whole-droplet chain).
// W: ring (droplet j cannot start before droplet j-W has finished... approximated by finish time of j-W), 0 = unbounded.  returns makespan in steps
double dataflow(const uint32_t *cells, const int64_t *off, const double *steps, int64_t D, int NX, int shift, int whole, int64_t W, double *finish_out) {
	int nbb = (NX >> shift) + 1;
	double *avail = calloc((size_t)nbb*nbb, sizeof(double));
	double *fin = calloc(D, sizeof(double));
	double makespan = 0;
	for (int64_t j = 0; j < D; ++j) {
		int64_t n = off[j+1] - off[j];
		if (n == 0) {fin[j] = (j >= W && W) ? fin[j-W] : 0; continue;}
		const uint32_t *a = cells + off[j];
		double dt = steps[j]/(double)n;
		double cur = (W && j >= W) ? fin[j-W] : 0.0;
		if (whole) {
			for (int64_t k = 0; k < n; ++k) {uint32_t c = a[k] >> 1; int b = ((c / NX) >> shift)*nbb + ((c % NX) >> shift); if (avail[b] > cur) cur = avail[b];}
			cur += steps[j];
			for (int64_t k = 0; k < n; ++k) {if (a[k] & 1) {uint32_t c = a[k] >> 1; int b = ((c / NX) >> shift)*nbb + ((c % NX) >> shift); if (cur > avail[b]) avail[b] = cur;}}
		}
		else {
			for (int64_t k = 0; k < n; ++k) {
				uint32_t c = a[k] >> 1; int b = ((c / NX) >> shift)*nbb + ((c % NX) >> shift);
				if (avail[b] > cur) cur = avail[b];
				cur += dt;
				if (a[k] & 1) {avail[b] = cur;} // serial order: a later writer's time is >= (it waited for this cell)
			}
		}
		fin[j] = cur;
		if (cur > makespan) makespan = cur;
	}
	if (finish_out) memcpy(finish_out, fin, D*sizeof(double));
	free(avail); free(fin);
	return makespan;
}
