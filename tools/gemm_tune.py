"""Vendor-GEMM solution search (VERDICT r5 missing 6 / next-round item 3): the dense projections run hipBLASLt through torch, which asks the library's heuristic for ONE
solution per shape.  This tool has PyTorch TunableOp time EVERY hipBLASLt solution that is valid for each GEMM the pair-step issues, and measures what the winners buy.

  python tools/gemm_tune.py tune  [--configs cfg2,cfg5,cfg5bf16] [--out videogpa_amd/tuned/tunableop_gfx950.csv]
        runs `bench.py --layers 2 --steps 1` per config with TunableOp tuning ON (a 2-block model issues every per-layer GEMM of the full one: same M, N, K, strides)
        and merges the results into ONE file.  ~1 minute per distinct large shape.
  python tools/gemm_tune.py table [--file ...] [--json gpurun_out/gemm_tune_table.json]
        for every tuned entry with >= 1e11 FLOP: rebuilds the operands from the entry's key (m, n, k, lda, ldb, ldc), then times the library default and the tuned solution
        back to back in this process -- ms AND joules per launch (rsmi energy counter, tools/energy.py) -- and prints the table.
  python tools/gemm_tune.py prune [--table gpurun_out/gemm_tune_table.json] [--min-gain 1.005]
        keeps only the entries whose tuned solution the table confirms (faster AND not more joules): what ships in videogpa_amd/tuned/.
The step-level A/B (what counts) is `python bench.py --no-tuned-gemms` against `python bench.py` in one session: kernels["hipblaslt_gemm (vendor)"] and energy.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
DEFAULT_FILE = os.path.join(ROOT, "videogpa_amd", "tuned", "tunableop_gfx950.csv")
BENCH_ARGS = {"cfg2": ["--config", "cfg2"], "cfg3": ["--config", "cfg3"], "cfg4": ["--config", "cfg4"], "cfg5": ["--config", "cfg5"], "cfg5bf16": ["--config", "cfg5", "--no-fp8"]}


def tune(args):
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    merged, header = {}, []
    if os.path.isfile(args.out) and not args.fresh:
        header, merged = _read(args.out)
    for name in args.configs.split(","):
        tmp = os.path.join(ROOT, "gpurun_out", f"tunableop_{name}.csv")
        os.makedirs(os.path.dirname(tmp), exist_ok=True)
        if os.path.exists(tmp):
            os.remove(tmp)
        env = dict(os.environ, PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME=tmp, PYTORCH_TUNABLEOP_ROCBLAS_ENABLED="0",
                   PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=str(args.ms), PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="5", PYTORCH_TUNABLEOP_VERBOSE="1")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *BENCH_ARGS[name], "--layers", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-other-configs",
               "--no-kernel-timer"]
        t0 = time.perf_counter()
        with open(os.path.join(ROOT, "gpurun_out", f"tunableop_{name}.log"), "w") as log:
            r = subprocess.run(cmd, env=env, stdout=log, stderr=subprocess.STDOUT, timeout=args.timeout)
        print(f"{name}: exit {r.returncode} in {time.perf_counter() - t0:.0f} s", flush=True)
        if not os.path.isfile(tmp) and os.path.isfile(tmp[:-4] + "0.csv"):      # some torch versions insert the device ordinal
            tmp = tmp[:-4] + "0.csv"
        if os.path.isfile(tmp):
            h, rows = _read(tmp)
            header = h or header
            merged.update(rows)
            print(f"{name}: {len(rows)} entries", flush=True)
    with open(args.out, "w") as f:
        for ln in header:
            f.write(ln + "\n")
        for (op, key), (sol, ms) in sorted(merged.items()):
            f.write(f"{op},{key},{sol},{ms}\n")
    print(f"wrote {args.out}: {len(merged)} entries")


def _read(path):
    header, rows = [], {}
    with open(path) as f:
        for ln in f:
            ln = ln.rstrip("\n")
            if not ln:
                continue
            if ln.startswith("Validator"):
                header.append(ln)
                continue
            parts = ln.split(",")
            if len(parts) >= 4:
                rows[(parts[0], parts[1])] = (parts[2], parts[3])
    return header, rows


def _parse_key(key):
    """tn_<m>_<n>_<k>_ld_<lda>_<ldb>_<ldc>  (column-major BLAS: y[M,N] = x[M,K] W[N,K]^T  <->  m = N, n = M, k = K, lda = W row stride, ldb = x row stride, ldc = y row stride)"""
    p = key.split("_")
    if len(p) < 8 or p[4] != "ld":
        return None
    return {"trans": p[0], "m": int(p[1]), "n": int(p[2]), "k": int(p[3]), "lda": int(p[5]), "ldb": int(p[6]), "ldc": int(p[7])}


def table(args):
    import torch
    import torch.cuda.tunable as tn
    import torch.nn.functional as F
    from energy import read_joules
    header, rows = _read(args.file)
    dev = "cuda"
    out_rows = []
    tn.enable(False)
    for (op, key), (sol, ms_rec) in sorted(rows.items()):
        k = _parse_key(key)
        if k is None or not op.endswith("_TN") or "BFloat16" not in op:
            continue
        N, M, K = k["m"], k["n"], k["k"]
        fl = 2.0 * M * N * K
        if fl < args.min_flop:
            continue
        if op.startswith("ScaledGemm"):
            # the e4m3 feed-forward GEMMs of cfg5 (ops.frozen_linear_fp8: row-wise scales, bf16 out): torch._scaled_mm(x8 [M,K], w8 [N,K]^T, scale_a [M,1], scale_b [1,N])
            if "Float8_e4m3fn_Float8_e4m3fn" not in op or "_rw_1_" not in key or k["lda"] != K or k["ldb"] != K:
                continue
            x8 = torch.randn(M, K, device=dev).clamp(-3, 3).to(torch.float8_e4m3fn)
            w8 = torch.randn(N, K, device=dev).clamp(-3, 3).to(torch.float8_e4m3fn)
            sa, sb = torch.rand(M, 1, device=dev) + 0.5, torch.rand(1, N, device=dev) + 0.5
            bias = torch.randn(N, device=dev).bfloat16() if key.endswith("bias_BFloat16") else None
            fn = lambda: torch._scaled_mm(x8, w8.t(), scale_a=sa, scale_b=sb, bias=bias, out_dtype=torch.bfloat16)
            W = x = None
        else:
            W = (torch.randn(N, k["lda"], device=dev) / K ** 0.5).bfloat16()[:, :K]
            x = torch.randn(M, k["ldb"], device=dev).bfloat16()[:, :K]
            bias = torch.randn(N, device=dev).bfloat16() if op.startswith("GemmAndBias") else None
            fn = lambda: F.linear(x, W, bias)

        def measure():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            n = max(8, int(args.seconds / max(time.perf_counter() - t0, 1e-4)))
            e0, t0 = read_joules(), time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            t1, e1 = time.perf_counter(), read_joules()
            J = (e1 - e0) / n if (e0 is not None and e1 is not None) else float("nan")
            return (t1 - t0) / n * 1e3, J
        res = {}
        for leg in ("default", "tuned", "default2", "tuned2"):          # interleaved twice: the part's clock drifts over a session
            tuned = leg.startswith("tuned")
            tn.enable(tuned)
            if tuned:
                tn.tuning_enable(False)
                tn.set_filename(args.file, False)
            res[leg] = measure()
        tn.enable(False)
        d_ms, t_ms = (res["default"][0] + res["default2"][0]) / 2, (res["tuned"][0] + res["tuned2"][0]) / 2
        d_J, t_J = (res["default"][1] + res["default2"][1]) / 2, (res["tuned"][1] + res["tuned2"][1]) / 2
        row = {"op": op, "key": key, "M": M, "N": N, "K": K, "lda_W": k["lda"], "ld_x": k["ldb"], "bias": bias is not None, "solution": sol, "default_ms": d_ms, "tuned_ms": t_ms,
               "default_J": d_J, "tuned_J": t_J, "speedup": d_ms / t_ms, "default_tflops": fl / d_ms / 1e9, "tuned_tflops": fl / t_ms / 1e9,
               "default_tflop_per_J": fl / d_J / 1e12, "tuned_tflop_per_J": fl / t_J / 1e12}
        out_rows.append(row)
        print(f"M={M:6d} N={N:6d} K={K:6d} ldW={k['lda']:6d} ldx={k['ldb']:6d} bias={int(bias is not None)} {sol:28s} default {d_ms:7.3f} ms {d_J:6.3f} J {fl / d_ms / 1e9:7.1f} TF/s | "
              f"tuned {t_ms:7.3f} ms {t_J:6.3f} J {fl / t_ms / 1e9:7.1f} TF/s | x{d_ms / t_ms:5.3f}", flush=True)
        del W, x
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        with open(args.json, "w") as f:
            json.dump({"file": os.path.relpath(args.file, ROOT), "validators": header, "rows": out_rows}, f, indent=1)


def prune(args):
    """keep only the entries the steady-state table (ms AND joules per launch at the part's power cap, default and tuned interleaved) confirms: TunableOp times every
    candidate in a short burst at a clock the step never sees (its FF2 time: 1.51 ms; the same solution back to back: 2.0 ms), so some of its winners LOSE under
    the cap.  Entries without a table row (small GEMMs) go back to the library's heuristic too."""
    header, rows = _read(args.file)
    tab = json.load(open(args.table))["rows"]
    keep = {}
    for r in tab:
        key = r.get("key") or f"tn_{r['N']}_{r['M']}_{r['K']}_ld_{r['lda_W']}_{r['ld_x']}_{r['N']}"
        gain_t, gain_j = r["default_ms"] / r["tuned_ms"], r["default_J"] / r["tuned_J"]
        ok = r["solution"] != "Default" and gain_t >= args.min_gain and gain_j >= 1.0
        print(f"{r['op']:36s} {key:44s} {r['solution']:26s} time x{gain_t:5.3f} joules x{gain_j:5.3f} -> {'KEEP' if ok else 'default heuristic'}")
        if ok:
            keep[(r["op"], key)] = rows[(r["op"], key)]
    with open(args.out, "w") as f:
        for ln in header:
            f.write(ln + "\n")
        for (op, key), (sol, ms) in sorted(keep.items()):
            f.write(f"{op},{key},{sol},{ms}\n")
    print(f"wrote {args.out}: {len(keep)} of {len(rows)} entries kept")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("tune")
    a.add_argument("--configs", default="cfg2")
    a.add_argument("--out", default=DEFAULT_FILE)
    a.add_argument("--ms", type=int, default=30, help="tuning time per candidate solution")
    a.add_argument("--timeout", type=int, default=2400)
    a.add_argument("--fresh", action="store_true", help="do not merge into an existing file")
    b = sub.add_parser("table")
    b.add_argument("--file", default=DEFAULT_FILE)
    b.add_argument("--json", default=os.path.join(ROOT, "gpurun_out", "gemm_tune_table.json"))
    b.add_argument("--min-flop", type=float, default=1e11)
    b.add_argument("--seconds", type=float, default=1.0)
    c = sub.add_parser("prune")
    c.add_argument("--file", default=DEFAULT_FILE)
    c.add_argument("--table", default=os.path.join(ROOT, "gpurun_out", "gemm_tune_table.json"))
    c.add_argument("--out", default=DEFAULT_FILE)
    c.add_argument("--min-gain", type=float, default=1.005)
    args = ap.parse_args()
    {"tune": tune, "table": table, "prune": prune}[args.cmd](args)
