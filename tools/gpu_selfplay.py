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
groups = [int(x) for x in os.environ.get("SP_GROUPS", "2,3,4").split(",")]
host_api.set_ladder_budget(int(os.environ.get("SP_LADDER_BUDGET", "0")))   # 0 = reference-exact
for thr in ([threads] if len(sys.argv) > 3 else [16]):
    for ng in groups:
        host_api.set_groups(ng)
        st = host_api.selfplay_run(path, 1024 * ng, thr, secs)
        print(f"hip-engine groups={ng} games={1024*ng} threads={thr}: {st.positions/st.seconds:,.0f} pos/s  "
              f"moves/s={st.moves/st.seconds:,.0f} batches={st.batches} run_ms={st.gpu_seconds/max(st.batches,1)*1e3:.2f} "
              f"host_busy={st.host_seconds/st.seconds:.2f} finished_games={st.games} ladder={host_api.ladder_stats()}", flush=True)
