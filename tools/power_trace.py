"""Socket power and shader clock while a command runs -- the evidence behind DESIGN section 4.2's "the matrix pipe is power-limited":
    python tools/power_trace.py --out gpurun_out/power_TAG.json -- python bench.py --steps 6 --warmup 1 --no-cpu-baseline
Polls `rocm-smi --showpower --showclocks --showmaxpower --json` (about 3 samples per second) until the command exits and writes every
sample plus a summary over the BUSY samples (power above half of the largest one seen): mean / max socket power against the board's power
cap, mean / min shader clock against its top level."""
import argparse
import json
import re
import subprocess
import sys
import time


def _num(s):
    m = re.search(r"[-+]?\d+(\.\d+)?", str(s))
    return float(m.group(0)) if m else None


def sample():
    try:
        raw = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        card = next(iter(json.loads(raw).values()))
    except Exception as e:                                  # noqa: BLE001  (a missed sample is not an error)
        return {"error": str(e)}
    out = {"t": time.time()}
    for k, v in card.items():
        kl = k.lower()
        if "power" in kl and "max" in kl:
            out["power_cap_w"] = _num(v)
        elif "power" in kl and ("socket" in kl or "average" in kl or "current" in kl):
            out["power_w"] = _num(v)
        elif kl.startswith("sclk clock speed") or "sclk clock speed" in kl:
            out["sclk_mhz"] = _num(v)
        elif "mclk clock speed" in kl:
            out["mclk_mhz"] = _num(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    if not cmd:
        sys.exit("power_trace.py: no command given")
    proc = subprocess.Popen(cmd)
    samples = []
    while proc.poll() is None:
        samples.append(sample())
        time.sleep(0.1)
    good = [s for s in samples if "power_w" in s and s["power_w"] is not None]
    summ = {"command": " ".join(cmd), "returncode": proc.returncode, "samples": len(samples), "samples_with_power": len(good)}
    if good:
        top = max(s["power_w"] for s in good)
        busy = [s for s in good if s["power_w"] >= 0.5 * top]
        pw = [s["power_w"] for s in busy]
        ck = [s["sclk_mhz"] for s in busy if s.get("sclk_mhz")]
        summ.update(busy_samples=len(busy), power_w_mean=sum(pw) / len(pw), power_w_max=top, power_cap_w=next((s["power_cap_w"] for s in good if s.get("power_cap_w")), None))
        if ck:
            summ.update(sclk_mhz_mean=sum(ck) / len(ck), sclk_mhz_min=min(ck), sclk_mhz_max=max(ck))
    json.dump({"summary": summ, "samples": samples}, open(a.out, "w"), indent=1)
    print(json.dumps(summ))
    sys.exit(proc.returncode)


if __name__ == "__main__":
    main()
