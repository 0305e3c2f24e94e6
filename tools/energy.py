"""Socket energy counter of GPU 0 (librocm_smi64 through ctypes; `rocm-smi --showenergycounter` as the fall-back reader).

    e0 = read_joules(); <work>; torch.cuda.synchronize(); e1 = read_joules()     ->  e1 - e0 joules over the interval

The counter is the accumulated-energy register the driver exposes (resolution 15.26 uJ on MI300-class parts), sampled by the SMU about once per
millisecond: bracket at least a second of back-to-back launches.  Measurement helper only (joules per launch)."""
import ctypes
import re
import subprocess

_lib = None


def _load():
    global _lib
    if _lib is None:
        for name in ("librocm_smi64.so", "/opt/rocm/lib/librocm_smi64.so", "librocm_smi64.so.1"):
            try:
                lib = ctypes.CDLL(name)
            except OSError:
                continue
            if lib.rsmi_init(ctypes.c_uint64(0)) == 0:
                _lib = lib
                break
        if _lib is None:
            _lib = False
    return _lib


def read_joules(device=0):
    lib = _load()
    if lib:
        cnt, res, ts = ctypes.c_uint64(0), ctypes.c_float(0.0), ctypes.c_uint64(0)
        if lib.rsmi_dev_energy_count_get(ctypes.c_uint32(device), ctypes.byref(cnt), ctypes.byref(res), ctypes.byref(ts)) == 0:
            return cnt.value * float(res.value) * 1e-6
    out = subprocess.run(["rocm-smi", "-d", str(device), "--showenergycounter"], capture_output=True, text=True, timeout=20).stdout
    m = re.search(r"Accumulated Energy \(uJ\):\s*([0-9.]+)", out)
    if not m:
        raise RuntimeError("no energy counter: " + out[-300:])
    return float(m.group(1)) * 1e-6


if __name__ == "__main__":
    import time
    a = read_joules()
    time.sleep(1.0)
    b = read_joules()
    print(f"idle: {b - a:.1f} J over 1 s")
