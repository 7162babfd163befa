"""Self-play throughput on the GPU box: positions/s with the HIP engine behind the host."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from p3achygo_amd import host_api, netspec
model = sys.argv[1] if len(sys.argv) > 1 else "b12c256btl3"
games = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
secs = float(sys.argv[4]) if len(sys.argv) > 4 else 10
cfg = netspec.CONFIGS[model]
path = os.path.join(tempfile.mkdtemp(), model + ".p3w")
netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
print("cpus", len(os.sched_getaffinity(0)))
for thr in ([threads] if len(sys.argv) > 3 else [8, 16]):
    st = host_api.selfplay_run(None, games, thr, 3.0)
    print(f"null-engine  threads={thr}: {st.positions/st.seconds:,.0f} pos/s  batches={st.batches}")
    st = host_api.selfplay_run(path, games, thr, secs)
    print(f"hip-engine   threads={thr}: {st.positions/st.seconds:,.0f} pos/s  moves/s={st.moves/st.seconds:,.0f} "
          f"batches={st.batches} gpu_busy={st.gpu_seconds/st.seconds/2:.2f} host_busy={st.host_seconds/st.seconds:.2f} games={st.games}")
