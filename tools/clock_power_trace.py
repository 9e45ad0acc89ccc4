#!/usr/bin/env python3
"""Engine clock and socket power while the deblocking kernel runs (ON the GPU box).

As a module: `Sampler(device_index)` reads the amdgpu sysfs files of the card whose PCI address HIP reports for that device
(pp_dpm_sclk / pp_dpm_mclk / pp_dpm_fclk, hwmon power1_input / power1_average, temperatures) from a background thread;
bench.py wraps its timed region in one and reports the medians (`roofline.engine_clock_MHz`, `roofline.socket_power_W`).

As a script: starts bench.py with many timed launches as a child process, samples every 20 ms and prints the medians over
the child's timed window plus a thinned trace:
   python3 tools/clock_power_trace.py [bench.py arguments, e.g. --variant copy]"""
import glob, json, os, subprocess, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rd(p):
    try:
        with open(p) as fh:
            return fh.read().strip()
    except OSError:
        return None


def _sources():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if _rd(os.path.join(card, "vendor")) != "0x1002":
            continue
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "gpu_busy_percent"):
            p = os.path.join(card, name)
            if os.path.exists(p):
                out.setdefault(card, {})[name] = p
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for f in ("power1_average", "power1_input", "power1_cap", "temp1_input", "temp2_input", "temp3_input"):
                p = os.path.join(hw, f)
                if os.path.exists(p):
                    out.setdefault(card, {})[f] = p
    return out


def _current_mhz(txt):
    """pp_dpm_* lists the levels, the active one marked '*': '1: 2400Mhz *'"""
    if not txt:
        return None
    for line in txt.splitlines():
        if line.rstrip().endswith("*"):
            try:
                return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            except (IndexError, ValueError):
                return None
    return None


def pci_of_hip_device(index):
    """'0000:0a:00.0' of HIP device `index` (initialises the HIP runtime in this process), or None.  Script use only:
    bench.py hands Sampler the id of its open context (hevcdbk_device_pci_bus_id) instead."""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
            return buf.value.decode().lower()
    except OSError:
        pass
    return None


class Sampler:
    """Background reader of one card's clock / power files.  start() ... stop() -> dict of medians (None where unreadable)."""

    def __init__(self, device_index=0, period_s=0.01, pci=None):
        self.pci = pci or pci_of_hip_device(device_index)
        src = _sources()
        mine = [c for c in sorted(src) if self.pci and os.path.basename(os.path.realpath(c)).lower() == self.pci]
        self.card = mine[0] if mine else None
        self.files = src.get(self.card, {}) if self.card else {}
        self.period = period_s
        self.rows = []
        self._stop = threading.Event()
        self._thr = None

    def read_once(self):
        r = {"t": time.monotonic()}  # CLOCK_MONOTONIC: the clock hevcdbk_device_replay stamps t_begin / t_end with
        for k, p in self.files.items():
            v = _rd(p)
            if k.startswith("pp_dpm"):
                r[k] = _current_mhz(v)
            else:
                try:
                    r[k] = float(v)
                except (TypeError, ValueError):
                    r[k] = None
        return r

    def _run(self):
        while not self._stop.is_set():
            self.rows.append(self.read_once())
            self._stop.wait(self.period)

    def start(self):
        if self.files:
            self.rows = []
            self._stop.clear()
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()
        return self

    def stop(self, t_begin=None, t_end=None):
        if self._thr is not None:
            self._stop.set()
            self._thr.join()
            self._thr = None
        rows = [r for r in self.rows if (t_begin is None or r["t"] >= t_begin) and (t_end is None or r["t"] <= t_end)]
        return self.summary(rows)

    def summary(self, rows):
        def med(k):
            v = sorted(x[k] for x in rows if x.get(k) is not None)
            return v[len(v) // 2] if v else None
        power = med("power1_input") if "power1_input" in self.files else med("power1_average")
        cap = _rd(self.files["power1_cap"]) if "power1_cap" in self.files else None
        return {"card": self.card, "pci": self.pci, "samples": len(rows),
                "engine_clock_MHz": med("pp_dpm_sclk"), "memory_clock_MHz": med("pp_dpm_mclk"), "fabric_clock_MHz": med("pp_dpm_fclk"),
                "socket_power_W": None if power is None else power / 1e6,
                "power_cap_W": None if cap in (None, "") else float(cap) / 1e6,
                "temps_C": [None if med(k) is None else med(k) / 1e3 for k in ("temp1_input", "temp2_input", "temp3_input") if k in self.files]}


def main():
    smp = Sampler(0, period_s=0.02)
    if not smp.files:
        print(json.dumps({"error": "no amdgpu sysfs files readable for HIP device 0", "pci": smp.pci}))
        return
    steps = 4000
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--no-cpu-baseline", "--no-e2e", "--no-extra",
           "--traffic", "none"] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    smp.start()
    child.wait()
    smp.stop()
    rows = smp.rows
    out = child.stdout.read().strip().splitlines()
    bench = json.loads(out[-1]) if out else {}
    busy = [r for r in rows if (r.get("gpu_busy_percent") or 0) >= 90] or rows
    dur = bench.get("ms_per_step", 0.8) * steps / 1e3   # the timed window is the last steps * ms of the busy period
    tend = busy[-1]["t"]
    win = [r for r in busy if r["t"] >= tend - dur * 0.9]
    res = smp.summary(win)
    res.update({"window_s": dur, "kernel_avg_ms": bench.get("roofline", {}).get("kernel_avg_ms"),
                "frac": bench.get("roofline", {}).get("frac"), "variant": bench.get("config", {}).get("kernel_variant"),
                "bench_args": sys.argv[1:],
                "sclk_trace_MHz_every_10th_sample": [x.get("pp_dpm_sclk") for x in rows[::10]],
                "power_trace_W_every_10th_sample": [round((x.get("power1_input") or x.get("power1_average") or 0) / 1e6) for x in rows[::10]]})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
