"""Developer tool (lives under tests/ because it uses the oracle as the checker / timed CPU reference): throughput of batched motion-only BA."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slslam_amd import capi, synth
from oracle import pyoracle as O
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
base = [synth.make_motion_only(900 + i, num_lines=150) for i in range(64)]
ws = [base[i % 64] for i in range(nb)]
b = capi.LBABatch()
for w in ws: b.add(w)
b.finalize()
b.solve(); b.download(); b.iterations(clear=True)
t = time.perf_counter()
for _ in range(5): b.reset(); b.solve()
its = b.iterations(); dt = time.perf_counter() - t
print("motion-only: %d frames x 150 lines: %.2f ms per batch, %.0f LM it/s, %.0f frames/s" % (nb, 1e3 * dt / 5, its / dt, 5 * nb / dt))
t = time.perf_counter(); n = 0
for w in base[:32]:
    x, s, _ = O.lba_solve(w); n += s["num_successful_steps"] + s["num_unsuccessful_steps"]
dt = time.perf_counter() - t
print("oracle 1 thread: %.0f LM it/s, %.0f frames/s" % (n / dt, 32 / dt))
