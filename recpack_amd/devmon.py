"""Clock and power of the GPU while a benchmark runs (bench.py: `device.sclk_mhz_under_load`, `device.power_w_under_load`).

The boxes of a pool differ: the same binary ran the 100 000-item solve in 3.03 - 3.23 s in round 4, and the chip lowers its clock
under an MFMA-dense load (MI355X_MICROARCH.md, 'DVFS give-back').  A sampler thread reads the amdgpu hwmon files of the device
(`freq1_input` = shader clock in Hz, `power1_average` / `power1_input` in microwatts) every `period` seconds - plain sysfs reads, no
subprocess, nothing on the GPU - so that a bench line carries the clock its numbers were measured at.  The guide's caveat applies:
the in-kernel clock of an MFMA loop can read up to ~10 % below the DPM value reported here; it is evidence for "which box",
not a cycle count.  Falls back to `rocm-smi --json` (one subprocess per sample) when the hwmon files are not there.
"""
from __future__ import annotations

import glob
import json
import os
import subprocess
import threading
import time


def _hwmon_dirs():
    """[(pci address of the card, hwmon directory)] of every amdgpu card that reports a shader clock"""
    out = []
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*"), key=lambda c: int(c.rsplit("card", 1)[1])):
        try:
            pci = os.path.basename(os.path.realpath(os.path.join(card, "device"))).lower()
        except OSError:
            pci = ""
        for h in sorted(glob.glob(os.path.join(card, "device", "hwmon", "hwmon*"))):
            if os.path.exists(os.path.join(h, "freq1_input")):
                out.append((pci, h))
    return out


def pci_address_of(device_index: int):
    """'dddd:bb:dd.f' of a HIP device (through torch's device properties; None when torch cannot say)"""
    try:
        import torch

        p = torch.cuda.get_device_properties(device_index)
        return f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    except Exception:
        return None


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


class DeviceMonitor:
    """with DeviceMonitor(local_device) as mon: ...; mon.summary() -> {"sclk_mhz": {...}, "power_w": {...}, "samples": n}"""

    def __init__(self, device_index: int = 0, period: float = 0.1):
        self.period = period
        self.dir = None
        dirs = _hwmon_dirs()
        pci = pci_address_of(device_index)
        self.matched = "none"
        if dirs:
            # the sysfs card of THIS HIP device: by PCI address (a leased box sees one device but sysfs lists every card of the host)
            hit = [h for a, h in dirs if pci and a == pci.lower()]
            if hit:
                self.dir, self.matched = hit[0], f"pci {pci}"
            else:
                self.dir, self.matched = dirs[min(device_index, len(dirs) - 1)][1], f"index {device_index} of {len(dirs)} cards (pci {pci} not found)"
        self.source = f"sysfs:{self.dir} ({self.matched})" if self.dir else "rocm-smi"
        self._stop = threading.Event()
        self._thread = None
        self.sclk, self.power = [], []
        self._marks = []  # (name, t0, t1) windows to summarise separately
        self._t = []

    def _sample(self):
        if self.dir:
            f = _read_int(os.path.join(self.dir, "freq1_input"))
            p = _read_int(os.path.join(self.dir, "power1_average"))
            if p is None:
                p = _read_int(os.path.join(self.dir, "power1_input"))
            return (f / 1e6 if f else None), (p / 1e6 if p else None)
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = d[sorted(d)[0]]
            sclk = power = None
            for k, v in card.items():
                if "sclk" in k.lower() and "(" in str(v):
                    sclk = float(str(v).split("(")[1].split("M")[0])
                if "power" in k.lower() and "(w)" in k.lower():
                    power = float(v)
            return sclk, power
        except Exception:
            return None, None

    def _run(self):
        while not self._stop.is_set():
            s, p = self._sample()
            self._t.append(time.perf_counter())
            self.sclk.append(s)
            self.power.append(p)
            self._stop.wait(self.period if self.dir else max(self.period, 0.5))

    def start(self):
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread:
            self._thread.join(timeout=10)

    __enter__ = start

    def __exit__(self, *exc):
        self.stop()

    @staticmethod
    def _stats(v):
        v = [x for x in v if x is not None]
        if not v:
            return None
        v = sorted(v)
        return {"min": round(v[0], 1), "mean": round(sum(v) / len(v), 1), "median": round(v[len(v) // 2], 1), "max": round(v[-1], 1)}

    def summary(self, t0: float | None = None, t1: float | None = None) -> dict:
        """Statistics over the samples taken in [t0, t1] (perf_counter times; all samples when omitted)."""
        idx = [i for i, t in enumerate(self._t) if (t0 is None or t >= t0) and (t1 is None or t <= t1)]
        return {
            "source": self.source,
            "samples": len(idx),
            "period_s": self.period,
            "sclk_mhz": self._stats([self.sclk[i] for i in idx]),
            "power_w": self._stats([self.power[i] for i in idx]),
        }
