#!/usr/bin/env python3
"""bench.py -- rows/sec through the infera_predict hot path on MI355X (BASELINE.json metric).

A "step" is ONE pass of the hot path over one batch of synthetic input: the whole 10M-row x
128-feature FLOAT table of BASELINE config C2 (3-layer MLP 128->256->64->1), already resident in
HBM when the timed region starts, pushed through the model that infera_load_model lowered.  With
--gpus N each rank owns its own 10M-row range of an (N x 10M)-row table (row-range sharding, weak
scaling, no data-path collective); value = all rows of all ranks / max-over-ranks time.

Launch: `python bench.py` (N=1) or
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch first: its bundled HIP runtime must be the one libinfera.so binds to (one runtime per process)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=None, help="table rows per GPU (default: C2 10M for mlp, C4 50M for logreg)")
    ap.add_argument("--workload", default="mlp", choices=["mlp", "logreg", "resnet18"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample duration")
    ap.add_argument("--share-device", type=int, default=None,
                    help="testing only: every rank uses this one HIP device (lets the N>1 control path run on a 1-GPU box)")
    return ap.parse_args()


def cpu_baseline(model_path: str, cols: int, target_s: float) -> dict:
    """The oracle ("port") timed on this box's host cores with the reference's execution shape:
    T threads, 2048-row chunks, per-cell boxed gather, single-threaded graph per chunk.  T is the best
    of a short sweep (containers often expose more logical CPUs than their cgroup lets them use at
    once; oversubscribed threads get throttled and the scan slows down)."""
    from oracle import oracle

    m = oracle.Model(model_path)
    ncpu = os.cpu_count() or 1
    try:
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    cands = sorted({max(1, ncpu // d) for d in (16, 8, 4, 2, 1)})
    best_t, best_rate = 1, 0.0
    for t in cands:
        rows = 2048 * t * 2
        sec, _ = m.bench_scan(rows, cols, seed=42, threads=t, chunk_rows=2048, boxed=True)
        if rows / sec > best_rate:
            best_t, best_rate = t, rows / sec
    rows = int(max(2048 * best_t * 2, min(best_rate * target_s, 50_000_000)) // 2048 * 2048)
    sec, _ = m.bench_scan(rows, cols, seed=42, threads=best_t, chunk_rows=2048, boxed=True)
    return {"value": rows / sec, "unit": "rows/s", "cores": best_t, "kind": "port",
            "sample": f"{rows} rows x {cols} f32 in 2048-row chunks, oracle/infera_oracle.c orc_bench_scan, "
                      f"boxed per-cell gather + single-threaded graph per chunk, {sec:.2f} s wall; "
                      f"threads = best of sweep {cands} on {os.cpu_count()} logical CPUs"}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = local_rank if args.share_device is None else args.share_device
    # One process per GPU: this rank's library instance must only create a context / upload weights on
    # ITS device (read once at library load, so set before importing the binding).
    os.environ.setdefault("INFERA_DEVICES", str(dev))
    if world > 1:
        # control plane only (barrier + max-reduce of one float); the data path has no collective
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(dev)
    torch.cuda.init()

    from infera_amd import capi, onnx_writer, shard

    if capi.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: " + capi.get_devices()["reason"])
    rows = args.rows or {"mlp": 10_000_000, "logreg": 50_000_000, "resnet18": 1024}[args.workload]
    hw = int(os.environ.get("INFERA_BENCH_RESNET_HW", "224"))  # experiments only; C5 is 224
    cols = 3 * hw * hw if args.workload == "resnet18" else 128
    tmp = tempfile.mkdtemp(prefix="infera_bench_")
    if args.workload == "mlp":
        path = onnx_writer.write(os.path.join(tmp, "mlp.onnx"), onnx_writer.mlp((128, 256, 64, 1)))
        out_cols, wl_name = 1, "C2: 3-layer MLP 128->256->64->1 (Gemm+Relu, Gemm+Relu, Gemm), 10M-row x 128-col FLOAT table"
        bound, flops_row, bytes_row = "mfma", 98432.0, 516.0
    elif args.workload == "resnet18":
        path = onnx_writer.write(os.path.join(tmp, "resnet18.onnx"), onnx_writer.resnet18(in_hw=hw))
        out_cols, wl_name = 1000, "C5: ResNet-18 topology (random weights), BLOB[3x224x224] f32 images resident in HBM"
        bound, flops_row, bytes_row = "mfma", 3628146688.0, 606112.0
    else:
        path = onnx_writer.write(os.path.join(tmp, "logreg.onnx"), onnx_writer.logreg_softmax(128, 10))
        out_cols, wl_name = 10, "C4: logistic regression Gemm(128->10)+Softmax(axis=1), 50M-row x 128-col FLOAT table, list output of 10"
        bound, flops_row, bytes_row = "hbm", 2560.0, 552.0
    capi.load_model("bench", path)
    plan = capi.get_plan("bench")

    d_in = capi.DeviceBuffer(dev, rows * cols * 4)
    d_out = capi.DeviceBuffer(dev, rows * out_cols * 4)
    row0, _ = shard.row_range(rank, world, rows)
    capi.synth_fill(d_in, 42, row0, rows, cols)  # this rank's row range of the global table

    def step():
        capi.predict_device("bench", d_in, rows, cols, d_out, sync=False)

    for _ in range(args.warmup):
        step()
    capi.sync(dev)

    barrier = shard.barrier
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    capi.sync(dev)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed)

    # roofline of the dominant kernel: HIP events on the launching stream around back-to-back launches
    iters = max(3, min(args.steps, 10))
    ms = capi.time_predict_device("bench", d_in, rows, cols, d_out, iters)
    kernel_s = ms / 1e3 / iters
    if bound == "mfma":
        achieved, peak, unit = flops_row * rows / kernel_s / 1e12, FP32_MFMA_PEAK_TFLOPS, "TFLOP/s"
    else:
        achieved, peak, unit = bytes_row * rows / kernel_s / 1e9, HBM_PEAK_GBS, "GB/s"

    # HBM traffic per launch: PMC counters cannot be read from inside this process; they are collected by
    # tools/profile_bench.sh (separate rocprofv3 --pmc passes of this same command) and committed as
    # profiles/traffic_<workload>.json.  Reported only when that file matches this workload and row count.
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("rows") == rows:
            traffic = tj["traffic_bytes_per_launch"]  # HBM bytes per launch, a plain number as the contract asks
            traffic_source = f"profiles/{os.path.basename(tpath)} (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, {tj.get('round', '')})"

    # a cheap end-of-run sanity check so a silently wrong kernel cannot post a number
    y = d_out.download((4, out_cols))
    assert os.environ.get("INFERA_CONV_PROBE") or all(map(lambda v: v == v, y.ravel().tolist())), "NaN in output"

    if rank == 0:
        total_rows = rows * world * args.steps
        line = {
            "metric": "rows/sec through infera_predict (device-resident table scan)",
            "value": total_rows / elapsed,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (counter-based splitmix64 table, seed 42; random-init weights seed 1234)",
            "config": {"workload": wl_name, "rows_per_gpu": rows, "features": cols, "parallelism": f"row-range x{world}",
                       "entry": "infera_hip_predict_device (inputs resident in HBM)",
                       "kernel": plan.get("fused_kernel", ",".join(plan["exec"]))},
            "roofline": {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes": bytes_row * rows,
                         "kernel_ms": kernel_s * 1e3, "algorithmic_per_row": {"flop": flops_row, "bytes": bytes_row}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(path, cols, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
