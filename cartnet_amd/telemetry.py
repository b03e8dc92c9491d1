"""Clock / power / temperature of the GPU a rank runs on, read from sysfs (plain file reads: no subprocess, no SMI
library, nothing that touches the HIP runtime).  bench.py samples it before and after its timed region so that a
number measured on a slow box can be told from a regression.

The card is found through its PCI address (``torch.cuda.get_device_properties(i)`` gives domain / bus / device), so the
sample belongs to the GPU the rank computes on even when sysfs shows all eight cards of the host.  Every field is
optional: a file that is missing or unreadable on a box becomes ``None`` and ``sources`` says what was found.
"""
from __future__ import annotations

import glob
import os
import re
import struct
from typing import Optional


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _read_int(path: str) -> Optional[int]:
    s = _read(path)
    try:
        return int(s) if s is not None else None
    except ValueError:
        return None


def pci_address(device_index: int = 0) -> Optional[str]:
    """'dddd:bb:dd.f' of a HIP device, or None.  Does not initialise the GPU more than torch already has."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        return f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
    except Exception:
        return None


_DIR_CACHE: dict = {}


def device_dir(device_index: int = 0) -> Optional[str]:
    """sysfs directory of the card: /sys/bus/pci/devices/<addr> if it matches, else the only amdgpu card there is."""
    if device_index in _DIR_CACHE:
        return _DIR_CACHE[device_index]
    _DIR_CACHE[device_index] = d = _device_dir(device_index)
    return d


def _device_dir(device_index: int) -> Optional[str]:
    addr = pci_address(device_index)
    if addr:
        d = os.path.join("/sys/bus/pci/devices", addr)
        if os.path.isdir(d):
            return d
    cards = [d for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
             if os.path.exists(os.path.join(d, "pp_dpm_sclk")) or glob.glob(os.path.join(d, "hwmon/hwmon*"))]
    if len(cards) == 1:
        return cards[0]
    if cards and 0 <= device_index < len(cards):
        return cards[device_index]           # best effort: same order as the runtime's enumeration
    return None


def _active_level_mhz(text: Optional[str]) -> Optional[int]:
    """pp_dpm_sclk / pp_dpm_mclk: lines like '1: 2400Mhz *' -- the starred one is current."""
    if not text:
        return None
    for line in text.splitlines():
        if "*" in line:
            m = re.search(r"(\d+)\s*[Mm][Hh]z", line)
            if m:
                return int(m.group(1))
    return None


def _gpu_metrics(d: str) -> dict:
    """The few fields of the binary `gpu_metrics` table whose offsets are the same in every v1.x layout: the header
    (size, format, content revision) and, for v1.4+ (MI300-class), temperature_hotspot / temperature_mem at 4 / 6,
    curr_socket_power at 18 is NOT stable across revisions, so only the header and the two temperatures are decoded."""
    out = {}
    try:
        with open(os.path.join(d, "gpu_metrics"), "rb") as f:
            raw = f.read(64)
    except OSError:
        return out
    if len(raw) >= 8:
        size, fmt, rev = struct.unpack_from("<HBB", raw, 0)
        out["gpu_metrics_rev"] = f"{fmt}.{rev}"
        if fmt == 1 and rev >= 4:
            hot, mem = struct.unpack_from("<HH", raw, 4)
            if 0 < hot < 200:
                out["hotspot_c_gpu_metrics"] = hot
            if 0 < mem < 200:
                out["mem_c_gpu_metrics"] = mem
    return out


def read(device_index: int = 0) -> dict:
    """One sample: {'sclk_mhz', 'mclk_mhz', 'power_w', 'power_cap_w', 'temp_c': {label: C}, 'sources': [...]}."""
    d = device_dir(device_index)
    out = {"sclk_mhz": None, "mclk_mhz": None, "power_w": None, "power_cap_w": None, "temp_c": {}, "sources": []}
    if d is None:
        out["sources"].append("no amdgpu sysfs directory found")
        return out
    out["sysfs"] = d
    sclk = _active_level_mhz(_read(os.path.join(d, "pp_dpm_sclk")))
    mclk = _active_level_mhz(_read(os.path.join(d, "pp_dpm_mclk")))
    if sclk is not None:
        out["sclk_mhz"] = sclk
        out["sources"].append("pp_dpm_sclk")
    if mclk is not None:
        out["mclk_mhz"] = mclk
        out["sources"].append("pp_dpm_mclk")
    for h in sorted(glob.glob(os.path.join(d, "hwmon/hwmon*"))):
        for name, key, scale in (("power1_average", "power_w", 1e-6), ("power1_input", "power_w", 1e-6),
                                 ("power1_cap", "power_cap_w", 1e-6)):
            v = _read_int(os.path.join(h, name))
            if v is not None and out[key] is None:
                out[key] = round(v * scale, 1)
                out["sources"].append("hwmon/" + name)
        for f in sorted(glob.glob(os.path.join(h, "temp*_input"))):
            v = _read_int(f)
            if v is None:
                continue
            label = _read(f.replace("_input", "_label")) or os.path.basename(f).replace("_input", "")
            out["temp_c"][label] = round(v / 1000.0, 1)
        for f in sorted(glob.glob(os.path.join(h, "freq*_input"))):
            v = _read_int(f)
            label = _read(f.replace("_input", "_label")) or ""
            if v is None:
                continue
            if label == "sclk" and out["sclk_mhz"] is None:
                out["sclk_mhz"] = int(v / 1e6)
                out["sources"].append("hwmon/" + os.path.basename(f))
            if label == "mclk" and out["mclk_mhz"] is None:
                out["mclk_mhz"] = int(v / 1e6)
                out["sources"].append("hwmon/" + os.path.basename(f))
    if out["temp_c"]:
        out["sources"].append("hwmon/temp*_input")
    out.update(_gpu_metrics(d))
    return out


def compact(sample: dict) -> dict:
    """The fields a bench line carries (no paths, no source list)."""
    t = sample.get("temp_c") or {}
    return {"sclk_mhz": sample.get("sclk_mhz"), "mclk_mhz": sample.get("mclk_mhz"), "power_w": sample.get("power_w"),
            "power_cap_w": sample.get("power_cap_w"),
            "edge_c": t.get("edge"), "junction_c": t.get("junction", sample.get("hotspot_c_gpu_metrics")),
            "mem_c": t.get("mem", sample.get("mem_c_gpu_metrics"))}


class Sampler:
    """Background thread that samples clock / power / temperature every `period` seconds while a stretch of GPU work
    runs; `stop()` returns mean / min / max of clock and power and the maximum temperatures -- what the card did UNDER
    the load, which a snapshot taken after the final synchronize cannot show.

    Kept light on purpose (the sampler shares the GIL with a ctypes-heavy main thread): the handful of sysfs
    files it needs (pp_dpm_sclk, pp_dpm_mclk, one hwmon power file, the junction and memory temperatures) are resolved
    ONCE here -- no glob, no gpu_metrics table, no label lookups per sample -- and every pass of bench.py uses the same
    period (>= 0.1 s by default).  ``bench.py --no-telemetry`` starts no sampler at all."""

    DEFAULT_PERIOD = 0.1

    def __init__(self, device_index: int = 0, period: Optional[float] = None):
        import threading
        self.device_index = device_index
        self.period = self.DEFAULT_PERIOD if period is None else max(float(period), 0.05)
        self.samples = []
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)
        d = device_dir(device_index)
        self._f_sclk = self._f_mclk = self._f_power = self._f_junction = self._f_mem = None
        if d is not None:
            f = os.path.join(d, "pp_dpm_sclk")
            self._f_sclk = f if os.path.exists(f) else None
            f = os.path.join(d, "pp_dpm_mclk")
            self._f_mclk = f if os.path.exists(f) else None
            for h in sorted(glob.glob(os.path.join(d, "hwmon/hwmon*"))):
                for name in ("power1_average", "power1_input"):
                    f = os.path.join(h, name)
                    if self._f_power is None and os.path.exists(f):
                        self._f_power = f
                for f in sorted(glob.glob(os.path.join(h, "temp*_input"))):
                    label = _read(f.replace("_input", "_label")) or ""
                    if label == "junction" and self._f_junction is None:
                        self._f_junction = f
                    if label == "mem" and self._f_mem is None:
                        self._f_mem = f

    def _sample(self) -> dict:
        p = _read_int(self._f_power) if self._f_power else None
        j = _read_int(self._f_junction) if self._f_junction else None
        m = _read_int(self._f_mem) if self._f_mem else None
        return {"sclk_mhz": _active_level_mhz(_read(self._f_sclk)) if self._f_sclk else None,
                "mclk_mhz": _active_level_mhz(_read(self._f_mclk)) if self._f_mclk else None,
                "power_w": round(p * 1e-6, 1) if p is not None else None,
                "junction_c": round(j / 1000.0, 1) if j is not None else None,
                "mem_c": round(m / 1000.0, 1) if m is not None else None}

    def _run(self):
        while not self._stop.is_set():
            self.samples.append(self._sample())
            self._stop.wait(self.period)

    def start(self) -> "Sampler":
        self._th.start()
        return self

    def stop(self) -> dict:
        self._stop.set()
        self._th.join()
        out = {"samples": len(self.samples), "period_s": self.period}
        for key in ("sclk_mhz", "mclk_mhz", "power_w"):
            v = [s[key] for s in self.samples if s.get(key) is not None]
            if v:
                out[key] = {"mean": round(sum(v) / len(v), 1), "min": min(v), "max": max(v)}
        for key in ("junction_c", "mem_c"):
            v = [s[key] for s in self.samples if s.get(key) is not None]
            if v:
                out[key + "_max"] = max(v)
        return out
