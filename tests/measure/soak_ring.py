"""Randomised soak of the asynchronous host-buffer entries (include/psm.h: psm_submit_grid_io / psm_wait_grid on pageable and on
registered memory, psm_ring_acquire / submit / wait) against the synchronous entry (run by hand on a GPU box; tests/test_ring.py
and tests/test_bound_guard.py hold the fixed cases).  Every trial draws a model, a grid shape, a case count, whether the
geometry is bound, a stream of inputs (velocities change from step to step; now and then ANOTHER obstacle arrives, which a bound
handle must notice on the device and answer on the general path), the entry, the number of tickets in flight and the order in
which they are waited for.  Without a geometry change every ticket must equal the synchronous solve of its input bit for bit;
with one, to the tolerance between the bound and the general path (2e-5 of the field's range) and with the same NaN pattern.

    python tests/measure/soak_ring.py [trials] [seed]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from psm_amd import GridSurrogate, synthetic

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
rng = np.random.default_rng(seed)
t0 = time.time()
n_bit = n_tol = n_trips = n_tickets = 0
worst = 0.0


def same(tag, got, want, exact, info):
    global worst, n_bit, n_tol
    if not np.array_equal(np.isnan(got), np.isnan(want)):
        raise SystemExit(f"NaN pattern differs: {tag} {info}")
    if exact:
        if not np.array_equal(got, want, equal_nan=True):
            raise SystemExit(f"not bit-identical to the synchronous solve: {tag} {info}: max diff {np.nanmax(np.abs(got - want)):.3e}")
        n_bit += 1
    else:
        ok = ~np.isnan(want)
        if ok.any():
            err = float(np.abs(got[ok] - want[ok]).max() / max(1e-6, np.abs(want[ok]).max()))
            worst = max(worst, err)
            if err > 2e-5:
                raise SystemExit(f"mismatch {err:.2e}: {tag} {info}")
        n_tol += 1


for trial in range(trials):
    variant = ("deltas", "gradp", "chapter5")[int(rng.integers(3))]
    ny, nx = int(rng.integers(128, 400)), int(rng.integers(256, 600))
    n_cases = int(rng.integers(1, 4))
    model = synthetic.make_model(variant, p_in=int(rng.integers(4, 100)), p_out=int(rng.integers(4, 100)), seed_pca=int(rng.integers(1 << 20)),
                                 seed_w=int(rng.integers(1 << 20)))

    def draw_geometry():
        return np.stack([synthetic.channel_grid(ny, nx, seed=int(rng.integers(1 << 30)), obstacle=("circle", "rectangle", "plate")[int(rng.integers(3))],
                                                cx=float(rng.uniform(0.2, 0.8)), cy=float(rng.uniform(0.25, 0.75)), r=float(rng.uniform(0.05, 0.18))).astype(np.float32)
                         for _ in range(n_cases)])
    base = draw_geometry()
    other = draw_geometry()
    bind = rng.random() < 0.7
    change = bind and rng.random() < 0.35                               # another obstacle somewhere in the stream
    n_steps = int(rng.integers(6, 20))
    k_change = int(rng.integers(1, n_steps)) if change else -1
    inputs, scales = [], []
    for k in range(n_steps):
        g = (other if k == k_change else base).copy()
        g[..., :model.sdf_ch] *= np.float32(rng.uniform(0.5, 1.5))
        inputs.append(g)
        scales.append([float(rng.uniform(0.3, 2.0)) for _ in range(n_cases)])
    mode = ("pageable", "registered", "zero_copy")[int(rng.integers(3))]
    depth = int(rng.integers(1, 7))
    fifo = rng.random() < 0.6
    info = dict(trial=trial, variant=variant, ny=ny, nx=nx, n_cases=n_cases, bind=bind, change=k_change, mode=mode, depth=depth, fifo=fifo, steps=n_steps)
    with GridSurrogate(model, ny, nx, max_cases=n_cases) as sur:
        bound = bool(bind and sur.bind_geometry(base))
        results = [None] * n_steps
        pend = []                                                       # (step, ticket, out view or None)
        gin = gout = None
        if mode == "registered":
            gin = np.ascontiguousarray(np.stack(inputs))
            gout = np.full((n_steps, n_cases, ny, nx, model.c_out), np.nan, np.float32)
            sur.host_register(gin); sur.host_register(gout)

        def retire(j):
            k, t, fo = pend.pop(j)
            if mode == "zero_copy":
                sur.ring_wait(t)
                results[k] = fo[:n_cases].copy()
            elif mode == "registered":
                sur.wait(t)
                results[k] = gout[k].copy()
            else:
                results[k] = sur.wait(t).copy()
        for k in range(n_steps):
            while len(pend) >= depth:
                retire(0 if fifo else int(rng.integers(len(pend))))
            while pend and k - min(p[0] for p in pend) >= 7:            # slots are handed out round-robin (8 of them): the oldest ticket
                retire(int(np.argmin([p[0] for p in pend])))            # must have been waited for before its slot comes round again
            if mode == "zero_copy":
                t, gi, fo = sur.ring_acquire()
                gi[:n_cases] = inputs[k]
                sur.ring_submit(t, n_cases, out_scale=scales[k])
                pend.append((k, t, fo))
            elif mode == "registered":
                pend.append((k, sur.submit(gin[k], out_scale=scales[k], out=gout[k]), None))
            else:
                pend.append((k, sur.submit(inputs[k], out_scale=scales[k]), None))
        while pend:
            retire(0 if fifo else int(rng.integers(len(pend))))
        if mode == "registered":
            sur.host_unregister(gin); sur.host_unregister(gout)
        trips = sur.guard_trips
        if bound and change and trips < 1:
            raise SystemExit(f"another obstacle went through a bound handle unnoticed: {info}")
        if (not change) and trips != 0:
            raise SystemExit(f"guard tripped on its own geometry: {info}")
        n_trips += trips
        # the synchronous entry on the same inputs, in the binding state the stream started in
        if bound and not sur.geometry_bound:
            assert sur.bind_geometry(base)
        for k in range(n_steps):
            if bound and k == k_change:
                sur.unbind_geometry()
            want = sur.solve(inputs[k], out_scale=scales[k])
            if bound and k == k_change:
                assert sur.bind_geometry(base)
            # bit-identical unless a geometry change dropped the binding somewhere in the stream (tickets behind it ran general)
            exact = not (bound and change)
            same(mode, results[k], want, exact, dict(info, step=k))
            n_tickets += 1
    if trial % 10 == 9:
        print(f"trial {trial + 1}/{trials}: {n_tickets} tickets, {n_bit} bit-identical, {n_tol} within 2e-5 (worst {worst:.2e}), {n_trips} guard trips, "
              f"{time.time() - t0:.0f} s", flush=True)
print(f"SOAK OK: {trials} streams, {n_tickets} tickets ({n_bit} bit-identical to the synchronous entry, {n_tol} behind a geometry change within 2e-5: worst "
      f"{worst:.2e}), {n_trips} guard trips, seed {seed}")
