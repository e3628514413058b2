#!/usr/bin/env python3
"""bench.py — Msamples/s decoded (16-bit stereo, 4096-frame packets) on N MI355X.

A "step" is one pass of the hot path (alacgpu_decode_batch_device: entropy decode -> predictor ->
unmix -> interleaved LE PCM) over one batch of synthetic packets that is already resident in HBM.
Weak scaling: every rank decodes its own --packets packets (default 65 536, the north-star batch);
ranks share nothing (no collective on the data path), `value` = samples decoded by all ranks per
second of the slowest rank. samples = frames x channels.

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (HBM, algorithmic
bytes / HIP-event kernel time) and, at N=1, `cpu_baseline` (the C oracle on the host cores).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--packets", type=int, default=65536, help="packets per GPU (weak scaling); BASELINE config e "
                    "(262 144 packets over 8 GPUs) is --gpus 8 --packets 32768")
    ap.add_argument("--depth", type=int, default=16)
    ap.add_argument("--channels", type=int, default=2)
    ap.add_argument("--frame-length", type=int, default=4096)
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gen-threads", type=int, default=0)
    ap.add_argument("--no-host-entry", action="store_true", help="skip the PCIe-inclusive leg (alacgpu_decode_batch)")
    ap.add_argument("--host-entry", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--profile", type=int, default=0, help="synth profile (0 = SURVEY 8d music model)")
    ap.add_argument("--out-stride-pad", type=int, default=0, help="extra bytes between PCM slots (experiments)")
    return ap.parse_args()


def kernels_of(dispatch):
    """The kernels of the last decode as the DEVICE dispatched them (alacgpu_last_dispatch reads the launch plan back):
    dominant kernel first."""
    parts = []
    if dispatch["narrow_slots"]:
        lanes = dispatch["lanes_per_packet"]
        parts.append("%s (%d narrow regular wave slots of %d packets, %d %s per CU%s)" % (
            dispatch["narrow_kernel"], dispatch["narrow_slots"], dispatch["packets_per_slot"], dispatch["workgroups_per_cu"],
            "wave pairs" if dispatch["gated"] else "four-wave workgroups",
            "; long predictors on %d lanes per packet" % lanes if lanes else ""))
    if dispatch["wide_slots"]:
        parts.append("%s (%d wide slots)" % (dispatch["wide_kernel"], dispatch["wide_slots"]))
    if dispatch["irregular_slots"]:
        parts.append("%s (%d irregular slots)" % (dispatch["irregular_kernels"], dispatch["irregular_slots"]))
    if len(parts) > 1 and dispatch["irregular_slots"] > dispatch["narrow_slots"] + dispatch["wide_slots"]:
        parts = parts[-1:] + parts[:-1]
    return " + ".join(parts)


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    return max(1, min(n, 64))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        # started by hand without the launcher: start it as a child before anything touches the GPU
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import numpy as np
    import torch
    import torch.distributed as dist

    pkg = importlib.import_module("saprobe-alac_amd")
    synth = importlib.import_module("saprobe-alac_amd.synth")
    if not os.path.exists(pkg.lib_path()):
        pkg.build()
    synth.build()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU decode path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run even a single rank joins the group
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev)

    P, FL, depth, ch = args.packets, args.frame_length, args.depth, args.channels
    cfg = pkg.PacketConfig(FrameLength=FL, BitDepth=depth, NumChannels=ch)
    bps = pkg.bytes_per_sample(depth)
    stride = FL * ch * bps + args.out_stride_pad

    # ---- synthetic packets of this rank's shard (seeded stream, packet index = rank*P + i) -------------
    t0 = time.time()
    threads = args.gen_threads or max(1, host_threads() // max(1, min(world, 8)))
    b = synth.gen_batch(cfg, P, profile=args.profile, first_index=rank * P, threads=threads)
    gen_s = time.time() - t0
    frames_total = int(b.frames.astype(np.int64).sum())
    samples = frames_total * ch
    alg_bytes = b.compressed_bytes + frames_total * ch * bps  # packet bytes in + PCM bytes out

    d_blob = torch.from_numpy(b.blob).to(dev)
    d_off = torch.from_numpy(b.offsets.astype(np.int64)).to(dev)
    d_sz = torch.from_numpy(b.sizes.astype(np.int32)).to(dev)
    d_out = torch.zeros((P, stride), dtype=torch.uint8, device=dev)
    d_fr = torch.zeros(P, dtype=torch.int32, device=dev)
    d_st = torch.full((P,), -1, dtype=torch.int32, device=dev)
    dec = pkg.NewPacketDecoder(cfg, local_rank)
    dec.reserve(P)
    torch.cuda.synchronize()  # the handle's stream does not order against torch's: uploads and fills must be done

    def step():
        dec.decode_batch_device(d_blob.data_ptr(), d_blob.numel(), d_off.data_ptr(), d_sz.data_ptr(), P, d_out.data_ptr(), stride,
                                d_fr.data_ptr(), d_st.data_ptr(), sync=False)

    def fence():
        dec.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    dec.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kms = dec.kernel_times_ms(min(args.steps, 64))
    kernel_ms = float(np.mean(kms)) if len(kms) else float("nan")
    dispatch = dec.last_dispatch()

    # ---- bit-exactness at full size: decode(encode(pcm)) == pcm, frame counts, status -------------------
    bit_exact = None
    if not args.no_verify:
        ok = int(d_st.abs().sum().item()) == 0
        ok = ok and bool(np.array_equal(d_fr.cpu().numpy().astype(np.uint32), b.frames))
        chunk = 8192
        for lo in range(0, P, chunk):
            exp = torch.from_numpy(b.pcm[lo:lo + chunk]).to(dev)
            ok = ok and bool(torch.equal(d_out[lo:lo + chunk, :exp.shape[1]], exp))
            del exp
        bit_exact = ok
        if use_dist:
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            bit_exact = bool(t.item())

    # ---- CPU baseline: the oracle (a C port of the reference algorithm) on the host cores, N=1 only -----
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle  # cpu_baseline leg only
        ocfg = oracle.make_config(FL, depth, ch)
        cores = host_threads()
        sample_n = min(P, 65536)
        t0 = time.perf_counter()
        _, cf, cs = oracle.decode_batch(ocfg, b.blob, b.offsets[:sample_n], b.sizes[:sample_n], threads=cores,
                                        want_output=False)
        cpu_s = time.perf_counter() - t0
        cpu_samples = int(cf.astype(np.int64).sum()) * ch
        t0 = time.perf_counter()
        one_n = min(sample_n, 4096)
        _, cf1, _ = oracle.decode_batch(ocfg, b.blob, b.offsets[:one_n], b.sizes[:one_n], threads=1,
                                        want_output=False)
        one_s = time.perf_counter() - t0
        cpu = {"value": round(cpu_samples / cpu_s / 1e6, 2), "unit": "Msamples/s", "cores": cores, "kind": "port",
               "sample": "first %d packets of the same batch, %d threads, static partition, %.2f s wall "
                         "(%.1f core-s)" % (sample_n, cores, cpu_s, cpu_s * cores),
               "single_thread_value": round(int(cf1.astype(np.int64).sum()) * ch / one_s / 1e6, 2),
               "all_ok": bool((cs == 0).all())}

    # ---- the host entry (alacgpu_decode_batch: dense blob -> chunked H2D / kernels / D2H on three streams), N=1 only;
    # PCIe-inclusive, never `value` ---------------------------------------------------------------------------------
    host_entry = None
    if world == 1 and rank == 0 and not args.no_host_entry:
        pk_off = np.zeros(P + 1, dtype=np.uint64)
        pk_off[1:] = np.cumsum(b.sizes.astype(np.uint64))
        nbytes = int(pk_off[-1])
        frame_bytes = FL * ch * bps

        def run_host(pinned):
            if pinned:  # memory the caller pinned itself is transferred in place
                t_blob = torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=True)
                t_out = torch.empty((P, frame_bytes), dtype=torch.uint8, pin_memory=True)
                t_fr = torch.empty(P, dtype=torch.int32, pin_memory=True)
                t_st = torch.empty(P, dtype=torch.int32, pin_memory=True)
                dense, h_out, h_fr, h_st = t_blob.numpy(), t_out.numpy(), t_fr.numpy().view(np.uint32), t_st.numpy()
            else:
                dense = np.empty(max(nbytes, 1), dtype=np.uint8)
                h_out = np.empty((P, frame_bytes), dtype=np.uint8)
                h_fr = np.empty(P, dtype=np.uint32)
                h_st = np.empty(P, dtype=np.int32)
            for i in range(P):  # the packets back to back, as an mdat holds them
                o = int(b.offsets[i])
                dense[int(pk_off[i]):int(pk_off[i + 1])] = b.blob[o:o + int(b.sizes[i])]
            best = None
            for _ in range(3):  # first call allocates the staging buffers
                h_st[:] = -1
                t0 = time.perf_counter()
                pkg._check(dec._lib.alacgpu_decode_batch(dec._h, dense.ctypes.data, dense.size, pk_off.ctypes.data, P,
                                                         h_out.ctypes.data, frame_bytes, h_fr.ctypes.data, h_st.ctypes.data))
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            ok = bool((h_st == 0).all()) and bool(np.array_equal(h_fr, b.frames))
            full = b.frames == FL  # a partial frame leaves the rest of its slot unspecified (decoder.go:127: output[:n])
            ok = ok and bool(np.array_equal(h_out[full], b.pcm[full]))
            for i in np.nonzero(~full)[0]:
                nb = int(b.frames[i]) * ch * bps
                ok = ok and bool(np.array_equal(h_out[i, :nb], b.pcm[i, :nb]))
            return {"value": round(samples / best / 1e6, 2), "unit": "Msamples/s", "seconds": round(best, 4),
                    "pcie_GBps": round((nbytes + frames_total * ch * bps) / best / 1e9, 2), "bit_exact": ok}

        host_entry = {"what": "alacgpu_decode_batch, dense host blob -> chunked H2D / kernels / D2H on three streams, "
                              "best of 3; bytes = packets up + PCM down",
                      "pageable": run_host(False), "pinned": run_host(True)}

    # HBM bytes per launch from the PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md): they cannot run
    # inside the bench, so the committed figure is used — but only if it was taken from THIS source of the kernels
    traffic, traffic_src, valu = None, None, None
    for tdir in sorted((d for d in os.listdir(os.path.join(ROOT, "profiles")) if d.startswith("r")), reverse=True):
        tpath = os.path.join(ROOT, "profiles", tdir, "traffic.json")
        if not os.path.exists(tpath):
            continue
        try:
            t = json.load(open(tpath))
            if t.get("csrc_sha256") == pkg.csrc_sha256() and t.get("workload") == [depth, ch, FL, P, args.profile]:
                traffic, traffic_src = t["traffic_bytes_per_launch"], "profiles/%s/traffic.json" % tdir
                valu = t.get("valu_insts_per_launch")
        except Exception:
            pass
        break
    # The resource the kernels are actually bound by: VALU issue. A wave64 VALU instruction occupies its SIMD-32 for two
    # cycles (MI355X_MICROARCH.md: Wave scheduling), so a launch of I wave-instructions cannot take fewer than
    # 2 I / (CUs x 4 SIMDs) cycles; frac = that floor / the cycles of the measured launch at the device's peak clock.
    props = torch.cuda.get_device_properties(local_rank)
    n_simd = props.multi_processor_count * 4
    clock_ghz = 2.4  # MI355X peak engine clock; the clock under load is lower, so frac understates the occupancy of the issue port
    valu_issue = None
    if valu and kernel_ms == kernel_ms:
        cycles = kernel_ms * 1e-3 * clock_ghz * 1e9
        valu_issue = {"insts_per_launch": int(valu), "cycles_per_inst": 2, "simds": n_simd, "clock_ghz": clock_ghz,
                      "cycles": int(cycles), "frac": round(2.0 * valu / n_simd / cycles, 4),
                      "source": traffic_src + " (SQ_INSTS_VALU, summed over the kernels of one decode)"}

    if rank == 0:
        value = samples * world * args.steps / elapsed / 1e6
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms == kernel_ms else None
        line = {
            "metric": "Msamples/s decoded (16-bit stereo, 4096-frame packets)" if (depth, ch, FL) == (16, 2, 4096)
            else "Msamples/s decoded (%d-bit %d-ch, %d-frame packets)" % (depth, ch, FL),
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%d packets/GPU x %d-bit %d-ch %d-frame ALAC, music-like synthetic, "
                                   "device-resident in/out" % (P, depth, ch, FL),
                       "packets_per_gpu": P, "samples_per_step_per_gpu": samples,
                       "compression_ratio": round(b.compressed_bytes / max(1, frames_total * ch * bps), 4),
                       "sharding": "independent packet ranges, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2) if achieved else None, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5) if achieved else None,
                         "traffic": traffic, "traffic_source": (traffic_src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload and this source)") if traffic else None,
                         "kernel": kernels_of(dispatch), "kernel_from": "alacgpu_last_dispatch: the launch plan read back from the device",
                         "dispatch": {k: dispatch[k] for k in ("slots", "irregular_slots", "wide_slots", "narrow_slots", "keys", "packets_per_slot", "gated", "lanes_per_packet", "workgroups_per_cu")},
                         "valu_issue": valu_issue,
                         "kernel_ms": round(kernel_ms, 4), "kernel_ms_is": "HIP events on the handle's stream around all kernels of one decode (sort pre-pass included)",
                         "algorithmic_bytes_per_launch": alg_bytes},
            "cpu_baseline": cpu, "bit_exact": bit_exact, "gen_seconds": round(gen_s, 2), "host_entry": host_entry,
        }
        if cpu:
            line["gpu_over_cpu"] = round(value / cpu["value"], 2)
        print(json.dumps(line), flush=True)
    dec.close()
    if use_dist:
        dist.destroy_process_group()
    if bit_exact is False:
        sys.exit(3)


if __name__ == "__main__":
    main()
