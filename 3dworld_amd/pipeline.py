"""Several heightmaps in flight on ONE GPU (the reference keeps eight generator objects in flight, height_gens[8], src/tiled_mesh.h:418): every map has its own terra
context (stream, scratch) and host thread and runs heightmap_t::proc_gen's device half -- noise + glaciate with the fused min(vals), then apply_erosion -- on it.

A map's erosion is a chain of dependent droplet steps on a few hundred waves; the NEXT map's noise kernel (vector-ALU bound, fills the chip) runs beside it.  Two noise kernels
at once only share the chip and finish together, so the maps take turns in their noise phase -- and the turn is handed over by the GPU itself: the thread of the next map waits
(terra_event_synchronize, on the host) for the event recorded right behind the previous map's noise kernel, not for the thread that launched that kernel to wake up, read
min(vals) back and release a semaphore (~65 us of the ~90 us between two noise kernels, profiles/r05_timeline_sparse_v2.txt).  min(vals) stays in device memory
(terra_gen_grid_minmax_async_dev -> terra_apply_erosion_devmin_dev): a map's erosion is enqueued directly behind its noise, no host round trip in between."""
import threading


class NoiseTurns:
    """the order in which the maps in flight enter their noise phase = the order in which their threads take a ticket here"""

    def __init__(self):
        self.lock = threading.Lock()
        self.last_ev, self.last_flag = None, None

    def take(self, my_ev):
        """-> (event of the noise before mine or None, flag that is set once that event has been recorded, my own flag to set after recording my_ev)"""
        mine = threading.Event()
        with self.lock:
            prev = (self.last_ev, self.last_flag)
            self.last_ev, self.last_flag = my_ev, mine
        return prev[0], prev[1], mine


def proc_gen_step(pkg, ctx, turns, ev, z_ptr, mm_ptr, x0, y0, dx, dy, nx, ny, droplets, flags=None, on_noise_enqueued=None):
    """one heightmap: wait for the noise turn, enqueue noise (+ min / max into mm_ptr: 2 device floats), hand the turn on, erode.  Returns when the map is complete in z_ptr."""
    prev_ev, prev_flag, my_flag = turns.take(ev) if turns is not None else (None, None, None)
    try:
        if prev_flag is not None:
            prev_flag.wait()             # the previous map's noise has been enqueued and its event recorded (long ago, in the steady state)
            ctx.event_synchronize(prev_ev)   # ... and has left the chip
        ctx.gen_grid_minmax_async_dev(z_ptr, x0, y0, dx, dy, nx, ny, mm_ptr, pkg.GEN_GLACIATE if flags is None else flags)
        if ev is not None:
            ctx.event_record(ev)
    finally:
        if my_flag is not None:
            my_flag.set()                # (also after a failure: the next map must not wait for a record that will never come)
    if on_noise_enqueued is not None:
        on_noise_enqueued()
    ctx.apply_erosion_devmin_dev(z_ptr, nx, ny, mm_ptr, droplets, pkg.ERODE_MINZ_IS_MIN)  # run_erosion passes min(vals): only written cells can need the clamp
