"""Shader clock and socket power of one GPU, sampled from a side thread (amdsmi; sysfs hwmon as the fall-back).
Measurement aid of bench.py and tools/gpu_clock_sample.py: the dominant kernel runs on the socket's power cap, and
the clock the chip grants under it is part of every timing here."""
import glob
import threading
import time


class PowerSampler:
    def __init__(self, device_index: int = 0):
        self.kind, self.h, self.cap_w = None, None, None
        self._samples, self._stop, self._th = [], False, None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.smi = amdsmi
            self.h = amdsmi.amdsmi_get_processor_handles()[device_index]
            self.read()
            self.kind = "amdsmi"
            try:
                cap = amdsmi.amdsmi_get_power_cap_info(self.h)
                c = cap.get("power_cap")
                if isinstance(c, (int, float)) and c > 0:
                    self.cap_w = float(c) / (1e6 if c > 1e5 else 1.0)   # microwatts on some versions
            except Exception:   # noqa: BLE001
                pass
        except Exception:   # noqa: BLE001
            self.kind = "sysfs"
            self.hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))

    def read(self):
        """(gfx clock MHz, socket power W) now."""
        if self.kind != "sysfs":
            s = self.smi
            clk = s.amdsmi_get_clock_info(self.h, s.AmdSmiClkType.GFX)
            pw = s.amdsmi_get_power_info(self.h)
            p = pw.get("current_socket_power", pw.get("average_socket_power"))
            return float(clk.get("clk", clk.get("cur_clk", 0))), float(p if isinstance(p, (int, float)) else 0)
        f = p = 0.0
        for h in self.hw[:1]:
            try:
                f = int(open(h + "/freq1_input").read()) / 1e6
                p = int(open(h + "/power1_average").read()) / 1e6
            except OSError:
                try:
                    p = int(open(h + "/power1_input").read()) / 1e6
                except OSError:
                    pass
        return f, p

    _RESIDENCIES = ("ppt", "socket_thm", "vr_thm", "hbm_thm", "prochot")

    def residencies(self):
        """The firmware's limiter residency accumulators (gpu_metrics): ticks of its accumulation counter during which the
        package power limit (ppt) / a thermal limit / PROCHOT held the clocks down; None when not reported."""
        if self.kind != "amdsmi":
            return None
        try:
            m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
            out = {"ticks": m["accumulation_counter"]}
            for k in self._RESIDENCIES:
                out[k] = m[k + "_residency_acc"]
            return out if all(isinstance(v, int) for v in out.values()) else None
        except Exception:   # noqa: BLE001
            return None

    def start(self, period_s: float = 0.02):
        self._samples, self._stop = [], False
        self._res0 = self.residencies()

        def loop():
            while not self._stop:
                try:
                    self._samples.append(self.read())
                except Exception:   # noqa: BLE001
                    pass
                time.sleep(period_s)

        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self):
        """Ends the sampling; returns a summary over the later three quarters of the samples (steady state) or None."""
        self._stop = True
        if self._th:
            self._th.join()
        res1 = self.residencies()
        s = [x for x in self._samples[len(self._samples) // 4:] if x[0] > 0 or x[1] > 0]
        if not s:
            return None
        clk = [x[0] for x in s]
        pw = [x[1] for x in s]
        out = {"sampler": self.kind, "samples": len(s), "gfx_clock_mhz_mean": sum(clk) / len(clk), "gfx_clock_mhz_min": min(clk),
               "gfx_clock_mhz_max": max(clk), "socket_power_w_mean": sum(pw) / len(pw), "socket_power_w_max": max(pw),
               "socket_power_cap_w": self.cap_w}
        r0 = getattr(self, "_res0", None)
        if r0 and res1 and res1["ticks"] > r0["ticks"]:
            dt = res1["ticks"] - r0["ticks"]
            # share of the sampled interval in which each limiter was holding the clocks down (firmware accumulators)
            out["limiter_residency"] = {k: (res1[k] - r0[k]) / dt for k in self._RESIDENCIES}
        return out
