/* tools/bench_native.c -- the headline loop of bench.py driven from plain C through include/terra.h (no Python, no torch): P heightmaps in flight,
 * one host thread and one context each, the noise turn handed over by GPU events.  A cross-check of the Python-driven number and an example of an engine-side caller.
 *   gcc -O2 -std=c99 -Iinclude tools/bench_native.c -L3dworld_amd -lterra_hip -lpthread -Wl,-rpath,$PWD/3dworld_amd -o tools/_bin/bench_native
 *   tools/_bin/bench_native [steps=64] [pipelines=4] [size=16384] [droplets=1000]                                                                  */
#define _POSIX_C_SOURCE 199309L
#include "terra.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {terra_ctx *ctx; float *z, *mm; terra_event *ev; terra_state st; int first, steps, stride, n, droplets, rc;} pipe_t;

/* The noise turn (one heightmap in its noise phase at a time, the others erode meanwhile) is handed over by the GPU: a thread takes a ticket = the event of the noise before
 * its own, waits on the host for that event (terra_event_synchronize), enqueues its noise + the event record, and erodes with min(vals) left in device memory -- the C form of
 * 3dworld_amd/pipeline.py (Pipeline). */
static pthread_mutex_t ticket_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t ticket_cv = PTHREAD_COND_INITIALIZER;
static terra_event *last_ev = NULL; /* the event behind the most recently ENQUEUED noise kernel */
static long tickets = 0, recorded = 0; /* tickets handed out / noise kernels enqueued so far (ticket k may wait on its predecessor's event once recorded >= k) */
static double now(void) {struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9*(double)t.tv_nsec;}
static void *worker(void *arg) {
	pipe_t *p = (pipe_t *)arg;
	for (int s = p->first; s < p->steps && p->rc == 0; s += p->stride) { /* heightmap_t::proc_gen on the device: noise + glaciate (+ fused min) -> erosion in place */
		pthread_mutex_lock(&ticket_lock);
		long const mine = tickets++;
		while (recorded < mine) {pthread_cond_wait(&ticket_cv, &ticket_lock);} /* the noise before mine has been enqueued and its event recorded (long ago, in the steady state) */
		terra_event *const prev = last_ev;
		pthread_mutex_unlock(&ticket_lock);
		if (prev) {p->rc = terra_event_synchronize(prev);}                     /* ... and has left the chip */
		if (p->rc == 0) {p->rc = terra_gen_grid_minmax_async_dev(p->ctx, -0.5f*(float)p->n, -0.5f*(float)p->n, p->st.DX_VAL, p->st.DY_VAL, (uint32_t)p->n, (uint32_t)p->n, TERRA_GEN_GLACIATE, 0, p->z, p->mm);}
		if (p->rc == 0) {p->rc = terra_event_record(p->ctx, p->ev);}
		pthread_mutex_lock(&ticket_lock); /* (also after a failure: the next ticket must not wait for a record that will never come) */
		last_ev = p->ev; recorded = mine + 1;
		pthread_cond_broadcast(&ticket_cv);
		pthread_mutex_unlock(&ticket_lock);
		if (p->rc == 0) {p->rc = terra_apply_erosion_devmin_dev(p->ctx, p->z, p->n, p->n, p->mm, (uint32_t)p->droplets, TERRA_ERODE_MINZ_IS_MIN);}
	}
	if (p->rc == 0) {p->rc = terra_synchronize(p->ctx);}
	return NULL;
}
static int run(pipe_t *pipes, int P, int steps) {
	pthread_t th[16];
	for (int i = 0; i < P; ++i) {pipes[i].steps = steps; pthread_create(&th[i], NULL, worker, &pipes[i]);}
	for (int i = 0; i < P; ++i) {pthread_join(th[i], NULL);}
	for (int i = 0; i < P; ++i) {if (pipes[i].rc) return pipes[i].rc;}
	return 0;
}
int main(int argc, char **argv) {
	int const steps = (argc > 1) ? atoi(argv[1]) : 64, P = (argc > 2) ? atoi(argv[2]) : 4, n = (argc > 3) ? atoi(argv[3]) : 16384, droplets = (argc > 4) ? atoi(argv[4]) : 1000;
	if (P < 1 || P > 16 || steps < 1 || n < 1) {fprintf(stderr, "bad arguments\n"); return 2;}
	if (terra_device_count() < 1) {fprintf(stderr, "no HIP device (there is no CPU fall-back)\n"); return 3;}
	terra_config c; memset(&c, 0, sizeof(c)); /* the synthetic scene of BASELINE.md section 3 (scene_config/config.txt:56-97), 8 octaves */
	c.mesh_x = c.mesh_y = 128; c.scene_x = c.scene_y = c.scene_z = 4.0f; c.mesh_height = 0.7f; c.mesh_scale = 1.0f;
	c.mesh_seed = 1; c.mesh_freq_filter = 1; c.mesh_gen_mode = TERRA_MGEN_SINE; c.mesh_gen_shape = 0; c.glaciate = 1;
	c.hmap[0] = 1000.0f; c.hmap[4] = 1000.0f; c.hmap[9] = 5.0f; c.hmap[10] = 0.001f; c.hmap[11] = -4.0f;
	c.erode_amount = 1.0f; c.start_mag = 0.02f; c.start_freq = 240.0f; c.mag_mult = 2.0f; c.freq_mult = 0.5f;
	pipe_t pipes[16]; memset(pipes, 0, sizeof(pipes));
	for (int i = 0; i < P; ++i) {
		pipe_t *p = &pipes[i];
		p->first = i; p->stride = P; p->n = n; p->droplets = droplets;
		if (terra_create(&p->ctx, 0) || terra_init_scene(p->ctx, &c) || terra_get_state(p->ctx, &p->st) || terra_malloc(p->ctx, (void **)&p->z, (size_t)n*(size_t)n*sizeof(float)) ||
		    terra_malloc(p->ctx, (void **)&p->mm, 2*sizeof(float)) || terra_event_create(p->ctx, &p->ev)) {
			fprintf(stderr, "setup failed: %s\n", terra_last_error()); return 1;
		}
	}
	if (run(pipes, P, 2*P)) {fprintf(stderr, "warm-up failed: %s\n", terra_last_error()); return 1;} /* scratch allocation, graph capture */
	double const t0 = now();
	if (run(pipes, P, steps)) {fprintf(stderr, "run failed: %s\n", terra_last_error()); return 1;}
	double const dt = now() - t0;
	printf("{\"driver\": \"C (include/terra.h)\", \"grid\": %d, \"droplets\": %d, \"pipelines\": %d, \"steps\": %d, \"ms_per_step\": %.4f, \"gcells_per_s\": %.2f}\n",
		n, droplets, P, steps, 1e3*dt/steps, (double)n*(double)n*steps/dt/1e9);
	for (int i = 0; i < P; ++i) {terra_event_destroy(pipes[i].ev); terra_free(pipes[i].ctx, pipes[i].mm); terra_free(pipes[i].ctx, pipes[i].z); terra_destroy(pipes[i].ctx);}
	return 0;
}
