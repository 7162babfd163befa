"""bench.py's side thread that samples shader clock and socket power (p3achygo_amd/power_sampler.py): without a GPU it
must stay out of the way — no exception, no summary."""


def test_sampler_without_a_gpu_reports_nothing():
    import time
    from p3achygo_amd.power_sampler import PowerSampler
    s = PowerSampler(0)
    assert s.kind in ("amdsmi", "sysfs")
    s.start(period_s=0.005)
    time.sleep(0.03)
    out = s.stop()
    if s.kind == "sysfs":
        assert out is None and s.residencies() is None      # this container: no hwmon of an AMD GPU, no amdsmi device
    else:
        assert out is None or out["samples"] >= 1


def test_sampler_summary_arithmetic():
    from p3achygo_amd.power_sampler import PowerSampler
    s = PowerSampler.__new__(PowerSampler)
    s.kind, s.cap_w, s._th, s._stop = "sysfs", 1400.0, None, True
    s._samples = [(0.0, 0.0)] * 4 + [(2100.0, 1390.0), (2200.0, 1400.0), (2000.0, 1380.0), (2100.0, 1390.0)] * 3
    out = s.stop()
    assert out["samples"] == 12 and out["gfx_clock_mhz_mean"] == 2100.0 and out["gfx_clock_mhz_max"] == 2200.0
    assert out["socket_power_w_mean"] == 1390.0 and out["socket_power_cap_w"] == 1400.0
    assert "limiter_residency" not in out
