"""Thread-per-game throughput: the reference's own host design (one OS thread per game blocking in
NNInterface::LoadAndGetInference, cc/nn/nn_interface.cc:108-133) over the HIP engine, to set
beside the non-blocking scheduler's number in bench.py.  Positions are random-legal playouts,
every position is evaluated once (cache off).

  python tools/gpu_nn_interface.py [threads=512] [moves=60] [net=b12c256btl3]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from p3achygo_amd import engine, host_api, netspec  # noqa: E402


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    moves = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    net = sys.argv[3] if len(sys.argv) > 3 else "b12c256btl3"
    import tempfile
    cfg = netspec.CONFIGS[net]
    path = os.path.join(tempfile.mkdtemp(), "n.p3w")
    netspec.save_p3w(path, cfg, netspec.generate_weights(cfg))
    L = host_api.lib()
    L.p3host_nn_new.restype = C.c_void_p
    L.p3host_nn_new.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_long, C.c_long, C.c_int,
                                C.c_char_p, C.c_int]
    L.p3host_nn_free.argtypes = [C.c_void_p]
    L.p3host_nn_num_inferences.restype = C.c_long
    L.p3host_nn_num_inferences.argtypes = [C.c_void_p]
    L.p3host_nn_play_threads_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_void_p]
    for strategy, name in ((0, "kMutex"), (1, "kGenCounter")):
        err = C.create_string_buffer(512)
        nn = L.p3host_nn_new(1, engine.LIB_PATH.encode(), path.encode(), 0, threads, 400, 0, strategy, err, 512)
        assert nn, err.value.decode()
        out = (engine.Result * (threads * moves))()
        L.p3host_nn_play_threads_ex(nn, threads, 4, 1, 0, C.byref(out))   # warm-up
        L.p3host_nn_free(nn)
        nn = L.p3host_nn_new(1, engine.LIB_PATH.encode(), path.encode(), 0, threads, 400, 0, strategy, err, 512)
        t0 = time.perf_counter()
        L.p3host_nn_play_threads_ex(nn, threads, moves, 1, 0, C.byref(out))
        dt = time.perf_counter() - t0
        infs = L.p3host_nn_num_inferences(nn)
        L.p3host_nn_free(nn)
        print(f"{name}: {threads} game threads x {moves} evaluations: {threads * moves / dt:,.0f} positions/s, "
              f"{infs} inferences, mean batch {threads * moves / infs:.0f}", flush=True)


if __name__ == "__main__":
    main()
