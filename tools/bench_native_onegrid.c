/* tools/bench_native_onegrid.c -- ONE heightmap per step on all GPUs of a node, one PROCESS per GPU, in plain C: include/terra.h + rccl.h, no Python, no torch.distributed.
 * The host-language form of 3dworld_amd/dist.py::OneHeightmapPipeline (heightmap_t::proc_gen, src/heightmap.cpp:130-187, on one shared grid):
 *   grid      `grids` terra_dgrids in flight: rank r's row strip lives in its own HBM, every rank maps all strips back to back (a strip crosses the process boundary as a
 *             file descriptor over a unix socket in a directory only this user can enter, SCM_RIGHTS, peer uid checked)
 *   step s    rank r: terra_gen_grid_rows_minmax_async_dev (its rows, {min, max} of the strip left in HBM) -> ncclAllReduce(min) of that one float ON THE SAME STREAM
 *             -> terra_event_record.  Rank s % ranks: an eroder context's stream waits for the event and runs terra_apply_erosion_devmin_dev over the mapped grid
 *             (min(vals) read from HBM by the final clamp; remote rows over xGMI).  Nothing of a step is read back by the host: the step's only host work is enqueueing.
 *   reuse     before a rank enqueues the all-reduce of step s it waits (host) for ITS erosion of step s - grids + 1: when that all-reduce completes anywhere, the grid
 *             that step s + 1 overwrites is final everywhere.
 *   bootstrap a shared-memory page made by the launching process: ncclUniqueId, barriers, the elapsed-time maximum.  --coll shm runs the per-step minimum through it as
 *             well (host round trip per step): RCCL refuses two ranks on one device, so that is how two ranks are tested on a one-GPU box (--same-device).
 *   --simulate-world W   ONE rank does what one rank of W does per step (1/W of the rows, the whole-grid erosion every W-th step, the all-reduce over a one-rank communicator):
 *             the per-rank step floor, measurable on one GPU.
 *   --shard-traces   the sparse erosion scheduler's read-only phases made by the strip owners (terra_erosion_shard_*): behind a step's all-reduce every rank traces the
 *             droplets that start in ITS rows into its own arena (one more terra_dgrid per grid in flight: the strips are the arenas) on a tracer context, a second one-float
 *             all-reduce on a communicator of its own says "all traces made", and the step's eroder gathers the traces through the mapping and checks / commits.
 *   --check   the last step's grid is compared byte for byte with the same map made by one context alone.
 * build: gcc -O2 -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tools/bench_native_onegrid.c -L3dworld_amd -lterra_hip -L/opt/rocm/lib -lrccl -lamdhip64 -lpthread \
 *            -Wl,-rpath,$PWD/3dworld_amd -Wl,-rpath,/opt/rocm/lib -o tools/_bin/bench_native_onegrid
 * usage: bench_native_onegrid [ranks=all GPUs] [steps=16] [size=16384] [droplets=1000] [--same-device] [--coll rccl|shm] [--grids 8] [--eroders 2] [--warmup 4]
 *                             [--simulate-world W] [--shard-traces] [--check]                                                                                                        */
#define _GNU_SOURCE
#include "terra.h"
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <signal.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#define MAXR 64
#define MAXG 32
#define MAXE 8
static double now(void) {struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9*(double)t.tv_nsec;}
static int g_rank = -1;
#define DIE(...) do {fprintf(stderr, "[rank %d] ", g_rank); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1);} while (0)
#define CK(x) do {if ((x) != 0) DIE("%s failed: %s", #x, terra_last_error());} while (0)
#define HCK(x) do {hipError_t e_ = (x); if (e_ != hipSuccess) DIE("%s failed: %s", #x, hipGetErrorString(e_));} while (0)
#define NCK(x) do {ncclResult_t r_ = (x); if (r_ != ncclSuccess) DIE("%s failed: %s", #x, ncclGetErrorString(r_));} while (0)

/* ---- the shared page: bootstrap and (--coll shm) the per-step minimum */
typedef struct {
	atomic_uint arrived[2];          /* all-reduce slots: counts up by `ranks` per use */
	float val[2][MAXR][2];
	atomic_uint bar;                 /* barrier: counts up by `ranks` per use */
	ncclUniqueId id, id2;
	double elapsed[MAXR];
} shared_t;
static shared_t *sh;
static int ranks = 1;
static unsigned bar_uses = 0, red_uses = 0;
static void spin_until(atomic_uint *c, unsigned target) {
	double const t0 = now();
	while ((int)(atomic_load_explicit(c, memory_order_acquire) - target) < 0) {
		if (now() - t0 > 120.0) DIE("a peer did not arrive within 120 s");
		sched_yield();
	}
}
static void shm_barrier(void) {++bar_uses; atomic_fetch_add_explicit(&sh->bar, 1u, memory_order_acq_rel); spin_until(&sh->bar, bar_uses*(unsigned)ranks);}
static void shm_allreduce_min2(float v[2]) { /* slot k & 1; a rank re-enters a slot only after everybody has left it (they all arrived in the other slot since) */
	unsigned const k = red_uses++, slot = k & 1u;
	sh->val[slot][g_rank][0] = v[0]; sh->val[slot][g_rank][1] = v[1];
	atomic_fetch_add_explicit(&sh->arrived[slot], 1u, memory_order_acq_rel);
	spin_until(&sh->arrived[slot], (k/2u + 1u)*(unsigned)ranks);
	for (int r = 0; r < ranks; ++r) {v[0] = fminf(v[0], sh->val[slot][r][0]); v[1] = fminf(v[1], sh->val[slot][r][1]);}
}

/* ---- descriptors of the strips between the ranks */
static void sock_path(struct sockaddr_un *a, char const *dir, int grid, int rank) {
	memset(a, 0, sizeof(*a)); a->sun_family = AF_UNIX;
	snprintf(a->sun_path, sizeof(a->sun_path), "%s/g%d_r%d.sock", dir, grid, rank);
}
static void exchange_fds(char const *dir, int grid, int fd, int *peer_fd /* [ranks], -1 = none */) {
	struct sockaddr_un me; sock_path(&me, dir, grid, g_rank);
	int const srv = socket(AF_UNIX, SOCK_STREAM, 0);
	if (srv < 0 || bind(srv, (struct sockaddr *)&me, sizeof(me)) != 0 || listen(srv, MAXR) != 0) DIE("unix socket %s: %s", me.sun_path, strerror(errno));
	shm_barrier(); /* everybody listens */
	for (int p = 0; p < ranks; ++p) { /* connects complete against the backlog, the message waits in the socket buffer: no thread needed */
		if (p == g_rank) continue;
		struct sockaddr_un to; sock_path(&to, dir, grid, p);
		int const c = socket(AF_UNIX, SOCK_STREAM, 0);
		if (c < 0 || connect(c, (struct sockaddr *)&to, sizeof(to)) != 0) DIE("connect %s: %s", to.sun_path, strerror(errno));
		int32_t who = g_rank;
		struct iovec iov = {&who, sizeof(who)};
		union {char buf[CMSG_SPACE(sizeof(int))]; struct cmsghdr align;} u; memset(&u, 0, sizeof(u));
		struct msghdr m; memset(&m, 0, sizeof(m)); m.msg_iov = &iov; m.msg_iovlen = 1; m.msg_control = u.buf; m.msg_controllen = sizeof(u.buf);
		struct cmsghdr *cm = CMSG_FIRSTHDR(&m); cm->cmsg_level = SOL_SOCKET; cm->cmsg_type = SCM_RIGHTS; cm->cmsg_len = CMSG_LEN(sizeof(int));
		memcpy(CMSG_DATA(cm), &fd, sizeof(int));
		if (sendmsg(c, &m, 0) != (ssize_t)sizeof(who)) DIE("sendmsg: %s", strerror(errno));
		close(c);
	}
	for (int n = 0; n < ranks - 1; ++n) {
		int const c = accept(srv, NULL, NULL);
		if (c < 0) DIE("accept: %s", strerror(errno));
		struct ucred cred; socklen_t cl = sizeof(cred);
		if (getsockopt(c, SOL_SOCKET, SO_PEERCRED, &cred, &cl) != 0 || cred.uid != getuid()) DIE("a descriptor message from another user: refused");
		int32_t who = -1;
		struct iovec iov = {&who, sizeof(who)};
		union {char buf[CMSG_SPACE(sizeof(int))]; struct cmsghdr align;} u; memset(&u, 0, sizeof(u));
		struct msghdr m; memset(&m, 0, sizeof(m)); m.msg_iov = &iov; m.msg_iovlen = 1; m.msg_control = u.buf; m.msg_controllen = sizeof(u.buf);
		if (recvmsg(c, &m, 0) != (ssize_t)sizeof(who)) DIE("recvmsg: %s", strerror(errno));
		struct cmsghdr *cm = CMSG_FIRSTHDR(&m);
		if (!cm || cm->cmsg_level != SOL_SOCKET || cm->cmsg_type != SCM_RIGHTS || cm->cmsg_len != CMSG_LEN(sizeof(int)) || who < 0 || who >= ranks || who == g_rank || peer_fd[who] >= 0) DIE("malformed descriptor message");
		memcpy(&peer_fd[who], CMSG_DATA(cm), sizeof(int));
		close(c);
	}
	close(srv); unlink(me.sun_path);
	shm_barrier();
}

/* ---- eroder threads: one context each, jobs in step order */
typedef struct {int s, g;} job_t;
typedef struct {
	terra_ctx *ctx; pthread_t th; pthread_mutex_t mu; pthread_cond_t cv;
	job_t *q; unsigned cap, head, tail; int stop; /* q: one entry per step of the run */
} eroder_t;
static eroder_t ero[MAXE];
static pthread_mutex_t done_mu = PTHREAD_MUTEX_INITIALIZER; static pthread_cond_t done_cv = PTHREAD_COND_INITIALIZER;
static unsigned char *done; /* [steps] */
static float *grid_ptr[MAXG]; static float *d_mm; static terra_event *ev_noise[MAXG];
static int N = 16384, droplets = 1000;
static int shard = 0; static char *arena_ptr[MAXG]; static size_t arena_stride = 0; static terra_event *ev_trace[MAXG]; static uint32_t row_end[MAXR]; /* --shard-traces */
static void *eroder_main(void *arg) {
	eroder_t *e = (eroder_t *)arg;
	for (;;) {
		pthread_mutex_lock(&e->mu);
		while (e->head == e->tail && !e->stop) {pthread_cond_wait(&e->cv, &e->mu);}
		if (e->head == e->tail) {pthread_mutex_unlock(&e->mu); return NULL;}
		job_t const j = e->q[e->head++ % e->cap];
		pthread_mutex_unlock(&e->mu);
		if (shard) { /* behind every rank's traces of the step; the traces come out of the ranks' arenas */
			CK(terra_event_wait(e->ctx, ev_trace[j.g]));
			CK(terra_erosion_shard_finish_dev(e->ctx, grid_ptr[j.g], N, N, d_mm + 2*j.g, (uint32_t)droplets, TERRA_ERODE_MINZ_IS_MIN, (uint32_t)ranks, (uint32_t)g_rank, row_end,
				arena_ptr[j.g] + (size_t)g_rank*arena_stride, arena_stride));
		}
		else {
		CK(terra_event_wait(e->ctx, ev_noise[j.g])); /* behind the step's noise and its all-reduce, on the device */
		CK(terra_apply_erosion_devmin_dev(e->ctx, grid_ptr[j.g], N, N, d_mm + 2*j.g, (uint32_t)droplets, TERRA_ERODE_MINZ_IS_MIN));
		}
		CK(terra_synchronize(e->ctx));
		pthread_mutex_lock(&done_mu); done[j.s] = 1; pthread_cond_broadcast(&done_cv); pthread_mutex_unlock(&done_mu);
	}
}
static void post(eroder_t *e, int s, int g) {pthread_mutex_lock(&e->mu); e->q[e->tail++ % e->cap] = (job_t){s, g}; pthread_cond_signal(&e->cv); pthread_mutex_unlock(&e->mu);}
static void wait_done(int s) {pthread_mutex_lock(&done_mu); while (!done[s]) {pthread_cond_wait(&done_cv, &done_mu);} pthread_mutex_unlock(&done_mu);}

static void scene(terra_config *c) { /* the synthetic scene of BASELINE.md section 3 (scene_config/config.txt:56-97), 8 octaves */
	memset(c, 0, sizeof(*c));
	c->mesh_x = c->mesh_y = 128; c->scene_x = c->scene_y = c->scene_z = 4.0f; c->mesh_height = 0.7f; c->mesh_scale = 1.0f;
	c->mesh_seed = 1; c->mesh_freq_filter = 1; c->mesh_gen_mode = TERRA_MGEN_SINE; c->mesh_gen_shape = 0; c->glaciate = 1;
	c->hmap[0] = 1000.0f; c->hmap[4] = 1000.0f; c->hmap[9] = 5.0f; c->hmap[10] = 0.001f; c->hmap[11] = -4.0f;
	c->erode_amount = 1.0f; c->start_mag = 0.02f; c->start_freq = 240.0f; c->mag_mult = 2.0f; c->freq_mult = 0.5f;
}
static size_t gcd_sz(size_t a, size_t b) {while (b) {size_t const t = a % b; a = b; b = t;} return a;}

typedef struct {int steps, warmup, same, use_rccl, G, E, sim, check, shard; char dir[64];} opts_t;

static int rank_main(opts_t const *o) {
	int const dev = o->same ? 0 : g_rank, world = ranks, sim = o->sim > 1 ? o->sim : 1;
	hipStream_t S = NULL;
	terra_config c; scene(&c);
	terra_ctx *nctx = NULL; terra_state st;
	CK(terra_create(&nctx, dev));
	if (o->use_rccl) { /* the noise context works on a stream of ours, so that the all-reduce can be enqueued between its kernels (--coll shm: no HIP or RCCL call in this file) */
		HCK(hipSetDevice(dev)); HCK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
		CK(terra_set_stream(nctx, (void *)S));
	}
	CK(terra_init_scene(nctx, &c)); CK(terra_get_state(nctx, &st));
	for (int i = 0; i < o->E; ++i) {
		CK(terra_create(&ero[i].ctx, dev)); CK(terra_init_scene(ero[i].ctx, &c));
		pthread_mutex_init(&ero[i].mu, NULL); pthread_cond_init(&ero[i].cv, NULL);
	}
	ncclComm_t comm = NULL, comm2 = NULL;
	terra_ctx *tctx = NULL; hipStream_t S2 = NULL; float *d_flag = NULL;
	shard = o->shard;
	if (shard) { /* the tracer context: this rank's probe / trace passes, behind the step's all-reduce, beside the next steps' noise */
		CK(terra_create(&tctx, dev));
		if (o->use_rccl) {HCK(hipStreamCreateWithFlags(&S2, hipStreamNonBlocking)); CK(terra_set_stream(tctx, (void *)S2));}
		CK(terra_init_scene(tctx, &c));
	}
	if (o->use_rccl) {
		if (g_rank == 0) {NCK(ncclGetUniqueId(&sh->id)); if (shard) {NCK(ncclGetUniqueId(&sh->id2));}}
		shm_barrier();
		ncclUniqueId id = sh->id;
		NCK(ncclCommInitRank(&comm, world, id, g_rank));
		if (shard) {ncclUniqueId id2 = sh->id2; NCK(ncclCommInitRank(&comm2, world, id2, g_rank));} /* "all traces made": never queued in front of the next steps' all-reduce(min) */
	}
	/* ---- the grids: strips whose byte size is a multiple of the mapping granularity */
	size_t const gran = terra_dgrid_granularity(nctx), row_bytes = (size_t)N*sizeof(float);
	if (gran == 0) DIE("no virtual memory management on this device: %s", terra_last_error());
	size_t const unit = gran/gcd_sz(gran, row_bytes), parts = (size_t)world*(size_t)sim;
	size_t per = ((size_t)N + parts - 1)/parts; per = (per + unit - 1)/unit*unit;
	/* (--simulate-world: one strip holds the whole grid, this rank fills the first 1/W of its rows per step) */
	size_t const strip_rows = (sim > 1) ? ((size_t)N + unit - 1)/unit*unit : per;
	size_t const r0 = (size_t)g_rank*per < (size_t)N ? (size_t)g_rank*per : (size_t)N, r1 = r0 + per < (size_t)N ? r0 + per : (size_t)N;
	if (r1 <= r0) DIE("no rows for this rank: fewer ranks or a larger grid");
	size_t strip_bytes[MAXR]; for (int r = 0; r < world; ++r) {strip_bytes[r] = strip_rows*row_bytes;}
	terra_dgrid *dg[MAXG];
	for (int g = 0; g < o->G; ++g) {
		CK(terra_dgrid_create(nctx, (uint32_t)world, strip_bytes, (uint32_t)g_rank, &dg[g]));
		if (world > 1) {
			int fd = -1, peer[MAXR]; for (int r = 0; r < world; ++r) {peer[r] = -1;}
			CK(terra_dgrid_export_fd(dg[g], &fd));
			exchange_fds(o->dir, g, fd, peer);
			close(fd);
			for (int r = 0; r < world; ++r) {if (r != g_rank) {CK(terra_dgrid_import_fd(dg[g], (uint32_t)r, peer[r])); close(peer[r]);}}
		}
		void *base = NULL; CK(terra_dgrid_map(dg[g], &base)); grid_ptr[g] = (float *)base;
		CK(terra_event_create(nctx, &ev_noise[g]));
	}
	terra_dgrid *ag[MAXG];
	if (shard) { /* the arenas: one more distributed array per grid in flight, strip r = rank r's arena */
		size_t const need = terra_erosion_shard_arena_bytes(tctx, (uint32_t)droplets);
		arena_stride = (need + gran - 1)/gran*gran;
		size_t ab[MAXR]; for (int r = 0; r < world; ++r) {ab[r] = arena_stride; row_end[r] = (uint32_t)(((size_t)(r + 1)*per < (size_t)N) ? (size_t)(r + 1)*per : (size_t)N);}
		for (int g = 0; g < o->G; ++g) {
			CK(terra_dgrid_create(nctx, (uint32_t)world, ab, (uint32_t)g_rank, &ag[g]));
			if (world > 1) {
				int fd = -1, peer[MAXR]; for (int r = 0; r < world; ++r) {peer[r] = -1;}
				CK(terra_dgrid_export_fd(ag[g], &fd));
				exchange_fds(o->dir, MAXG + g, fd, peer);
				close(fd);
				for (int r = 0; r < world; ++r) {if (r != g_rank) {CK(terra_dgrid_import_fd(ag[g], (uint32_t)r, peer[r])); close(peer[r]);}}
			}
			void *base = NULL; CK(terra_dgrid_map(ag[g], &base)); arena_ptr[g] = (char *)base;
			CK(terra_event_create(tctx, &ev_trace[g]));
		}
		if (o->use_rccl) {CK(terra_malloc(tctx, (void **)&d_flag, (size_t)o->G*sizeof(float))); HCK(hipMemset(d_flag, 0, (size_t)o->G*sizeof(float)));}
	}
	CK(terra_malloc(nctx, (void **)&d_mm, (size_t)o->G*2*sizeof(float)));
	int const total = o->warmup + o->steps;
	done = (unsigned char *)calloc((size_t)total + 1, 1);
	for (int i = 0; i < o->E; ++i) {
		ero[i].cap = (unsigned)total + 1u; ero[i].q = (job_t *)calloc(ero[i].cap, sizeof(job_t));
		if (!ero[i].q || pthread_create(&ero[i].th, NULL, eroder_main, &ero[i]) != 0) DIE("pthread_create");
	}
	float const x0 = -0.5f*(float)N, y0 = -0.5f*(float)N;
	int mine = 0, first = 0; double t0 = 0.0, elapsed = 0.0;
	for (int phase = 0; phase < 2; ++phase) { /* warm-up (scratch allocation, graph capture, RCCL's first launch), then the timed steps */
		int const s_end = phase == 0 ? o->warmup : total;
		for (int s = first; s < s_end; ++s) {
			int const g = s % o->G, j = s - o->G + 1;
			if (j >= 0 && (j % (world*sim)) == g_rank) {wait_done(j);} /* my erosion of the grid step s + 1 overwrites: complete before all_reduce(s) can complete anywhere */
			CK(terra_gen_grid_rows_minmax_async_dev(nctx, x0, y0, st.DX_VAL, st.DY_VAL, (uint32_t)N, (uint32_t)N, TERRA_GEN_GLACIATE, 0, (uint32_t)r0, (uint32_t)(r1 - r0),
				grid_ptr[g] + r0*(size_t)N, d_mm + 2*g));
			if (o->use_rccl) {NCK(ncclAllReduce(d_mm + 2*g, d_mm + 2*g, 1, ncclFloat, ncclMin, comm, S));}
			else if (world > 1) {
				float v[2] = {0.0f, 1.0f};
				CK(terra_memcpy_d2h(nctx, &v[0], d_mm + 2*g, sizeof(float)));
				shm_allreduce_min2(v);
				CK(terra_memcpy_h2d(nctx, d_mm + 2*g, &v[0], sizeof(float)));
			}
			CK(terra_event_record(nctx, ev_noise[g]));
			if (shard) { /* my strip's droplets, then "all traces made" */
				CK(terra_event_wait(tctx, ev_noise[g]));
				CK(terra_erosion_shard_trace_dev(tctx, grid_ptr[g], N, N, (uint32_t)droplets, (uint32_t)r0, (uint32_t)(r1 - r0), arena_ptr[g] + (size_t)g_rank*arena_stride));
				if (o->use_rccl) {NCK(ncclAllReduce(d_flag + g, d_flag + g, 1, ncclFloat, ncclMin, comm2, S2));}
				else if (world > 1) {float v[2] = {0.0f, 1.0f}; CK(terra_synchronize(tctx)); shm_allreduce_min2(v);}
				CK(terra_event_record(tctx, ev_trace[g]));
			}
			if ((s % (world*sim)) == g_rank) {post(&ero[mine++ % o->E], s, g);} else {pthread_mutex_lock(&done_mu); done[s] = 1; pthread_mutex_unlock(&done_mu);}
		}
		for (int s = first; s < s_end; ++s) {wait_done(s);} /* my erosions */
		CK(terra_synchronize(nctx));
		if (shard) {CK(terra_synchronize(tctx));}
		shm_barrier(); /* everybody's erosions: no grid is touched by anyone any more */
		if (phase == 0) {t0 = now();} else {elapsed = now() - t0;}
		first = s_end;
	}
	sh->elapsed[g_rank] = elapsed;
	shm_barrier();
	char const *verdict = "skipped";
	if (o->check && g_rank == 0 && sim == 1) { /* the last step's grid against the same map made by one context alone */
		size_t const cells = (size_t)N*(size_t)N;
		float *ref = NULL, mn = 0.0f, mx = 0.0f;
		float *ha = (float *)malloc(cells*sizeof(float)), *hb = (float *)malloc(cells*sizeof(float));
		if (!ha || !hb) DIE("--check: host memory");
		CK(terra_malloc(ero[0].ctx, (void **)&ref, cells*sizeof(float)));
		CK(terra_gen_grid_minmax_dev(ero[0].ctx, x0, y0, st.DX_VAL, st.DY_VAL, (uint32_t)N, (uint32_t)N, TERRA_GEN_GLACIATE, 0, ref, &mn, &mx));
		CK(terra_apply_erosion_dev(ero[0].ctx, ref, N, N, mn, (uint32_t)droplets, TERRA_ERODE_MINZ_IS_MIN));
		CK(terra_memcpy_d2h(ero[0].ctx, ha, ref, cells*sizeof(float)));
		CK(terra_memcpy_d2h(ero[0].ctx, hb, grid_ptr[(total - 1) % o->G], cells*sizeof(float)));
		verdict = memcmp(ha, hb, cells*sizeof(float)) == 0 ? "bit-equal" : "DIFFERENT";
		terra_free(ero[0].ctx, ref); free(ha); free(hb);
	}
	if (g_rank == 0) {
		double worst = 0.0; for (int r = 0; r < world; ++r) {worst = sh->elapsed[r] > worst ? sh->elapsed[r] : worst;}
		printf("{\"what\": \"one heightmap per step on all ranks (C driver, one process per GPU)\", \"ranks\": %d, \"same_device\": %d, \"simulate_world\": %d, \"coll\": \"%s\", \"grid\": %d, "
			"\"droplets\": %d, \"grids_in_flight\": %d, \"eroders\": %d, \"shard_traces\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.4f, \"%s\": %.2f, \"scaling\": \"strong\", \"check\": \"%s\"}\n",
			world, o->same ? 1 : 0, sim, o->use_rccl ? "rccl" : "shm", N, droplets, o->G, o->E, shard, o->steps, o->warmup, 1e3*worst/o->steps,
			sim > 1 ? "gcells_per_s_predicted_at_simulated_world" : "gcells_per_s", (double)N*(double)N*(double)o->steps/worst/1e9, verdict); /* a step is one whole map, whoever made which rows */
		fflush(stdout);
	}
	shm_barrier(); /* rank 0 has read the grids: nobody unmaps a strip a peer may still be reading */
	for (int i = 0; i < o->E; ++i) {pthread_mutex_lock(&ero[i].mu); ero[i].stop = 1; pthread_cond_signal(&ero[i].cv); pthread_mutex_unlock(&ero[i].mu); pthread_join(ero[i].th, NULL);}
	if (comm) {NCK(ncclCommDestroy(comm));}
	if (comm2) {NCK(ncclCommDestroy(comm2));}
	if (shard) {
		for (int g = 0; g < o->G; ++g) {terra_event_destroy(ev_trace[g]); terra_dgrid_destroy(ag[g]);}
		if (d_flag) {terra_free(tctx, d_flag);}
		terra_destroy(tctx);
		if (S2) {HCK(hipStreamDestroy(S2));}
	}
	for (int g = 0; g < o->G; ++g) {terra_event_destroy(ev_noise[g]); terra_dgrid_destroy(dg[g]);}
	terra_free(nctx, d_mm);
	for (int i = 0; i < o->E; ++i) {terra_destroy(ero[i].ctx);}
	terra_destroy(nctx);
	if (S) {HCK(hipStreamDestroy(S));}
	return strcmp(verdict, "DIFFERENT") == 0 ? 4 : 0;
}

int main(int argc, char **argv) {
	opts_t o; memset(&o, 0, sizeof(o)); o.steps = 16; o.warmup = 4; o.use_rccl = 1; o.G = 8; o.E = 2;
	int pos = 0, R = 0;
	for (int i = 1; i < argc; ++i) {
		if (strcmp(argv[i], "--same-device") == 0) {o.same = 1;}
		else if (strcmp(argv[i], "--check") == 0) {o.check = 1;}
		else if (strcmp(argv[i], "--shard-traces") == 0) {o.shard = 1;}
		else if (strcmp(argv[i], "--coll") == 0 && i + 1 < argc) {o.use_rccl = strcmp(argv[++i], "shm") != 0;}
		else if (strcmp(argv[i], "--grids") == 0 && i + 1 < argc) {o.G = atoi(argv[++i]);}
		else if (strcmp(argv[i], "--eroders") == 0 && i + 1 < argc) {o.E = atoi(argv[++i]);}
		else if (strcmp(argv[i], "--warmup") == 0 && i + 1 < argc) {o.warmup = atoi(argv[++i]);}
		else if (strcmp(argv[i], "--simulate-world") == 0 && i + 1 < argc) {o.sim = atoi(argv[++i]);}
		else {int const v = atoi(argv[i]); if (pos == 0) {R = v;} else if (pos == 1) {o.steps = v;} else if (pos == 2) {N = v;} else if (pos == 3) {droplets = v;} ++pos;}
	}
	if (o.same && R > 1 && o.use_rccl) {o.use_rccl = 0; o.same = 2;} /* RCCL refuses several ranks on one device (same = 2: say so once, in the launcher) */
	char const *child = getenv("TERRA_ONEGRID_RANK"); /* set by the launcher below: this process is one rank */
	if (child) {
		char const *dir = getenv("TERRA_ONEGRID_DIR");
		ranks = R; g_rank = atoi(child);
		if (!dir || R < 1 || g_rank < 0 || g_rank >= R) {fprintf(stderr, "not started by the launcher\n"); return 2;}
		snprintf(o.dir, sizeof(o.dir), "%s", dir);
		char path[96]; snprintf(path, sizeof(path), "%s/page", o.dir);
		int const fd = open(path, O_RDWR);
		if (fd < 0) {perror(path); return 2;}
		sh = (shared_t *)mmap(NULL, sizeof(shared_t), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		close(fd);
		if (sh == MAP_FAILED) {perror("mmap"); return 2;}
		int const rc = rank_main(&o);
		fflush(NULL);
		return rc;
	}
	/* ---- the launcher: it never touches the runtime, and every rank is a process of its own from exec on (what mpirun / torchrun would start) */
	if (R <= 0) { /* all GPUs: counted by a throw-away child */
		int p[2]; if (pipe(p) != 0) return 2;
		pid_t const c = fork();
		if (c == 0) {int n = terra_device_count(); if (write(p[1], &n, sizeof(n)) != (ssize_t)sizeof(n)) _exit(1); _exit(0);}
		int n = 0; if (read(p[0], &n, sizeof(n)) != (ssize_t)sizeof(n)) {n = 0;} waitpid(c, NULL, 0); close(p[0]); close(p[1]);
		R = n;
	}
	if (R < 1 || R > MAXR || o.steps < 1 || o.warmup < 0 || N < 256 || o.G < 2 || o.G > MAXG || o.E < 1 || o.E > MAXE || (o.sim > 1 && R != 1) || (o.sim > 1 && o.shard) || (o.shard && R > 16) || o.steps + o.warmup > 100000) {
		fprintf(stderr, "bad arguments (ranks %d)\n", R); return 2;
	}
	snprintf(o.dir, sizeof(o.dir), "/tmp/terra_onegrid_XXXXXX");
	if (!mkdtemp(o.dir)) {perror("mkdtemp"); return 2;} /* 0700: the shared page and the sockets of the descriptor exchange live here */
	char page[96]; snprintf(page, sizeof(page), "%s/page", o.dir);
	{
		int const fd = open(page, O_RDWR | O_CREAT | O_EXCL, 0600);
		if (fd < 0 || ftruncate(fd, (off_t)sizeof(shared_t)) != 0) {perror(page); return 2;} /* zero-filled */
		close(fd);
	}
	if (!getenv("NCCL_SOCKET_IFNAME")) {setenv("NCCL_SOCKET_IFNAME", "lo", 1);} /* the ranks of one node meet over the loopback interface: no name resolution */
	setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
	setenv("TERRA_ONEGRID_DIR", o.dir, 1);
	char rs[16], **av = (char **)calloc((size_t)argc + 2, sizeof(char *));
	if (!av) return 2;
	snprintf(rs, sizeof(rs), "%d", R);
	{ /* the ranks get the same arguments, with the rank count spelled out */
		int n = 0, seen = 0; av[n++] = argv[0];
		for (int i = 1; i < argc; ++i) {
			int const is_opt = strncmp(argv[i], "--", 2) == 0;
			if (is_opt) {av[n++] = argv[i]; if (strcmp(argv[i], "--same-device") != 0 && strcmp(argv[i], "--check") != 0 && strcmp(argv[i], "--shard-traces") != 0 && i + 1 < argc) {av[n++] = argv[++i];} continue;}
			av[n++] = (seen++ == 0) ? rs : argv[i];
		}
		if (seen == 0) {av[n++] = rs;}
		av[n] = NULL;
	}
	if (o.same == 2) {fprintf(stderr, "RCCL refuses several ranks on one device: --same-device runs with --coll shm\n");}
	pid_t pid[MAXR];
	for (int r = 0; r < R; ++r) {
		pid[r] = fork();
		if (pid[r] < 0) {perror("fork"); return 2;}
		if (pid[r] == 0) {char b[16]; snprintf(b, sizeof(b), "%d", r); setenv("TERRA_ONEGRID_RANK", b, 1); execv("/proc/self/exe", av); perror("execv"); _exit(127);}
	}
	int rc = 0, left = R;
	while (left > 0) { /* a rank that fails takes the others with it (they would wait for it in a collective) */
		int stv = 0; pid_t const p = wait(&stv);
		if (p < 0) break;
		--left;
		int const code = WIFEXITED(stv) ? WEXITSTATUS(stv) : 128 + WTERMSIG(stv);
		for (int r = 0; r < R; ++r) {if (pid[r] == p) {pid[r] = -1;}}
		if (code != 0 && rc == 0) {rc = code; for (int r = 0; r < R; ++r) {if (pid[r] > 0) {kill(pid[r], SIGTERM);}}}
	}
	unlink(page); rmdir(o.dir);
	return rc;
}
