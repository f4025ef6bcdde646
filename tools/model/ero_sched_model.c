// round-synchronous scheduler model on the serial-order access trace (block-level validation)
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
typedef struct {int32_t a, i;} dep_t;
static int64_t D; static int32_t *L;           // steps per droplet (groups of accesses; the last group of reads without move counts as a step too)
static int64_t *sofs;                           // per droplet: first global step index
static int64_t *dofs; static dep_t *deps;      // per global step: deps[dofs[g] .. dofs[g+1])
static int64_t nsteps_total;
// build: cells (cell<<1|w), off[D+1]; NX; shift (block = 1<<shift cells)
int64_t build(const uint32_t *cells, const int64_t *off, int64_t D_, int NX, int shift) {
	D = D_;
	int nbb = (NX >> shift) + 1; size_t nblk = (size_t)nbb*nbb;
	dep_t *lastw = malloc(nblk*sizeof(dep_t)); for (size_t b = 0; b < nblk; ++b) {lastw[b].a = -1; lastw[b].i = 0;}
	L = malloc(D*sizeof(int32_t)); sofs = malloc((D+1)*sizeof(int64_t));
	// pass 1: count steps
	int64_t g = 0;
	for (int64_t j = 0; j < D; ++j) {
		sofs[j] = g; int64_t n = off[j+1] - off[j]; const uint32_t *a = cells + off[j];
		int reads = 0; for (int64_t k = 0; k < n; ++k) {if (!(a[k] & 1)) ++reads;}
		int32_t l = (reads > 4) ? (reads - 4 + 3)/4 : 1; L[j] = l; g += l;
	}
	sofs[D] = g; nsteps_total = g;
	dofs = malloc((g+1)*sizeof(int64_t)); size_t dcap = (size_t)g*3 + 1024, dn = 0; deps = malloc(dcap*sizeof(dep_t));
	int32_t *wlast_blk = malloc(4096*sizeof(int32_t)); int32_t *wlast_step = malloc(4096*sizeof(int32_t));
	for (int64_t j = 0; j < D; ++j) {
		int64_t n = off[j+1] - off[j]; const uint32_t *a = cells + off[j];
		int reads = 0; int32_t step = 0; int nw = 0;
		int64_t gs = sofs[j]; int32_t cur_step = -1;
		for (int64_t k = 0; k < n; ++k) {
			uint32_t c = a[k] >> 1; int w = a[k] & 1;
			if (!w) {step = (reads < 4) ? 0 : (reads - 4)/4; ++reads;}
			while (cur_step < step) {++cur_step; dofs[gs + cur_step] = dn;}
			int32_t b = ((c / NX) >> shift)*nbb + ((c % NX) >> shift);
			dep_t lw = lastw[b];
			if (lw.a >= 0) { // dedupe within the step
				int dup = 0; for (size_t q = dofs[gs + cur_step]; q < dn; ++q) {if (deps[q].a == lw.a && deps[q].i == lw.i) {dup = 1; break;}}
				if (!dup) {if (dn == dcap) {dcap *= 2; deps = realloc(deps, dcap*sizeof(dep_t));} deps[dn++] = lw;}
			}
			if (w) {int f = -1; for (int q = 0; q < nw; ++q) {if (wlast_blk[q] == b) {f = q; break;}} if (f < 0) {if (nw < 4096) {f = nw++; wlast_blk[f] = b;}} if (f >= 0) wlast_step[f] = step;}
		}
		while (cur_step < L[j] - 1) {++cur_step; dofs[gs + cur_step] = dn;}
		for (int q = 0; q < nw; ++q) {lastw[wlast_blk[q]].a = (int32_t)j; lastw[wlast_blk[q]].i = wlast_step[q];}
	}
	dofs[g] = dn;
	free(lastw); free(wlast_blk); free(wlast_step);
	return g;
}
// simulate.  pipelined: deps resolve at step granularity (else whole droplet: dep (a,*) final only when a is fully final).  slice: steps per round for far droplets;
// near: the first `near` in-flight droplets run to the end (0: none).  ck: resume granularity (steps).  out[0] = rounds, out[1] = sum over rounds of the longest wave (steps), out[2] = total traced steps
void simulate(int64_t W, int pipelined, int slice, int near, int ck, double *out) {
	int32_t *F = calloc(D, sizeof(int32_t)), *Fn = calloc(D, sizeof(int32_t)), *P = calloc(D, sizeof(int32_t)); // F: final steps; P: position of the (speculative) trace
	int64_t base = 0; double rounds = 0, crit = 0, traced = 0;
	while (base < D) {
		int64_t hi = base + W < D ? base + W : D;
		int32_t longest = 0;
		for (int64_t j = base; j < hi; ++j) {
			Fn[j] = F[j];
			if (F[j] >= L[j]) continue;
			int budget = (near && j - base < near) ? 1 << 30 : slice;
			// the trace is at P[j] (>= F[j] rounded down to a checkpoint if it was invalidated); it runs `budget` steps
			int32_t start = P[j];
			int32_t end = start + budget < L[j] ? start + budget : L[j];
			int32_t ex = end - start; if (ex > longest) longest = ex; traced += ex;
			// which of the steps [F[j], end) become final: all inputs final at the start of the round
			int32_t t = F[j];
			if (start > t) {t = F[j];} // steps between F and start were traced earlier on data that was not final: they count as final only if their deps are final NOW and they were read...
			for (; t < end; ++t) {
				int64_t g = sofs[j] + t; int ok = 1;
				for (int64_t q = dofs[g]; q < dofs[g+1]; ++q) {
					dep_t d = deps[q];
					if (d.a < base) continue; // committed
					if (pipelined ? (F[d.a] > d.i) : (F[d.a] >= L[d.a])) continue;
					ok = 0; break;
				}
				if (!ok) break;
			}
			// steps in [F, start) traced in EARLIER rounds: their reads happened then; they are final only if the deps were final THEN.  Model: a step is final only if traced in a round
			// whose start had its deps final => steps before `start` that were not final stay not final until re-traced: the trace is pulled back to the checkpoint before the first such step
			if (t < start && F[j] < start) { // cannot happen: we always re-trace from <= first non-final step (below)
			}
			Fn[j] = t;
			// next round's trace position: continue from `end` if everything so far is final, else pull back to the checkpoint at or below the first non-final step
			P[j] = (t >= end) ? end : (t/ck)*ck;
			if (P[j] > end) P[j] = end;
		}
		for (int64_t j = base; j < hi; ++j) {F[j] = Fn[j];}
		while (base < D && F[base] >= L[base]) ++base;
		rounds += 1; crit += longest;
		if (rounds > 1e7) break;
	}
	out[0] = rounds; out[1] = crit; out[2] = traced;
	free(F); free(Fn); free(P);
}

// refined model: correctness of executed steps (speculation on visible versions pays when they turn out right).
// vis_mode 0: a droplet's version becomes visible when its trace completes (snapshot); 1: live partial (the running trace is visible as of the end of the previous round)
void simulate2(int64_t W, int vis_mode, int slice, int near, int ck, double *out) {
	int32_t *P = calloc(D, 4), *K = calloc(D, 4);       // current trace: position, leading correct steps
	int32_t *Kv = calloc(D, 4), *Pv = calloc(D, 4);     // visible version: leading correct steps, length (steps executed); Pv = 0: none
	int32_t *Kn = calloc(D, 4), *Pn = calloc(D, 4); uint8_t *done = calloc(D, 1), *pub = calloc(D, 1);
	int64_t base = 0; double rounds = 0, crit = 0, traced = 0;
	while (base < D) {
		int64_t hi = base + W < D ? base + W : D;
		int32_t longest = 0;
		for (int64_t j = base; j < hi; ++j) {
			Kn[j] = K[j]; Pn[j] = P[j]; pub[j] = 0;
			int complete = (P[j] >= L[j]);
			int32_t start = P[j];
			if (K[j] < P[j]) { // has incorrect steps: re-trace from the checkpoint before the first one, if that step's inputs are visible-correct now
				int64_t g = sofs[j] + K[j]; int ok = 1;
				for (int64_t q = dofs[g]; q < dofs[g+1]; ++q) {dep_t d = deps[q]; if (d.a < base) continue; if (Kv[d.a] > d.i) continue; ok = 0; break;}
				if (!ok) {if (complete) continue; /* a suspended incorrect trace keeps running below (it does not know) */}
				else {start = (K[j]/ck)*ck; complete = 0;}
			}
			else if (complete) continue;
			int budget = (near && j - base < near) ? 1 << 30 : slice;
			int32_t end = start + budget < L[j] ? start + budget : L[j];
			int32_t ex = end - start; if (ex > longest) longest = ex; traced += ex;
			int32_t k = (K[j] < start) ? K[j] : start; // correct prefix kept
			if (k == start) { // extend the correct prefix through the executed steps while inputs are visible-correct
				for (; k < end; ++k) {
					int64_t g = sofs[j] + k; int ok = 1;
					for (int64_t q = dofs[g]; q < dofs[g+1]; ++q) {dep_t d = deps[q]; if (d.a < base) continue; if (Kv[d.a] > d.i) continue; ok = 0; break;}
					if (!ok) break;
				}
			}
			Kn[j] = k; Pn[j] = end; pub[j] = 1;
		}
		for (int64_t j = base; j < hi; ++j) {
			K[j] = Kn[j]; P[j] = Pn[j];
			if (pub[j]) {
				if (vis_mode == 1 || P[j] >= L[j]) {Kv[j] = K[j]; Pv[j] = P[j];}
				else if (vis_mode == 0 && Pv[j] > 0 && K[j] < Kv[j]) {/* the old complete version stays visible; its correct prefix is what it was */}
			}
		}
		while (base < D && P[base] >= L[base] && K[base] >= L[base]) ++base;
		rounds += 1; crit += longest;
		if (rounds > 2e6) break;
	}
	out[0] = rounds; out[1] = crit; out[2] = traced;
	free(P); free(K); free(Kv); free(Pv); free(Kn); free(Pn); free(done); free(pub);
}
