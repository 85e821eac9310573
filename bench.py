#!/usr/bin/env python3
"""bench.py -- the reference's headline path on MI355X: mixed voices/sec at 48 kHz, 1024-sample
updates, HRTF stereo (BASELINE.json configs[2]: 4096 mono voices, bsinc24 resample, dual-ear
HRIR FIR, MixDirectHrtf post-process).  One "step" = one update of every voice.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0 (contract in the task statement; see DESIGN.md "Measurement").
Inputs (source PCM, voice state, the per-update parameter blocks of the moving voices) are
resident in HBM before the timed region starts.
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools", "measure"))      # oalmeasure: measurement loops over the public C-ABI

UPDATE_SAMPLES = 1024
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak, MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3         # fp32 dense peak (MFMA f32 = packed vector rate), MI355X_MICROARCH.md
# algorithmic flops per voice-update, SURVEY.md 8(d) / DESIGN.md 3.4:
#   config 3: bsinc24 resample 1024*24*4 + dual-ear FIR 1024*64*2*2 + gain 1024*4
#   config 2: bsinc24 resample + 5-line gain mix 1024*2*5
#   config 4: config 2 + on average 2 sends (4-line gain mix each, every third through a biquad pair)
#   config 5: config 3 + one send
FLOPS_PER_VOICE_UPDATE = {3: 1024 * 24 * 4 + 1024 * 64 * 2 * 2 + 1024 * 4, 2: 1024 * 24 * 4 + 1024 * 2 * 5,
                          4: 1024 * 24 * 4 + 1024 * 2 * 5 + 2 * (1024 * 2 * 4) + 2 * 1024 * 18 // 3,
                          5: 1024 * 24 * 4 + 1024 * 64 * 2 * 2 + 1024 * 4 + 1024 * 2 * 4}
# algorithmic bytes per voice-update, SURVEY.md 8(d) / DESIGN.md "Algorithmic bytes":
#   source window (941+48) f32 + mPrevSamples r/w + position r/w + HRTF history r/w + target HRIR
BYTES_PER_VOICE_UPDATE = {3: 3956 + 384 + 16 + 512 + 512, 2: 3956 + 384 + 16 + 2 * 4 * 5 + 4 * 5,
                          4: 3956 + 384 + 16 + 3 * 4 * 5 + 2 * 3 * 4 * 4, 5: 3956 + 384 + 16 + 512 + 512 + 3 * 4 * 4}


def build_scene(oalgpu, synth, api, config_id, nvoices, voice_base, mhr_bytes, vpg, num_real=None, voice_map=None, sample_fmt="f32"):
    """num_real: real output lines of a non-HRTF context (8 = a 7.1 device: the dry lines are then decoded
    to speaker feeds by the reference's X71 decoder in the post-process); None = no output stage.
    voice_map: the global voice index of every local voice (a shard dealt by cost class) instead of voice_base.
    sample_fmt: "f32" or "i16" source buffers (SURVEY.md 8d names both)."""
    hrtf = config_id in (3, 5)
    nsends = {4: 4, 5: 1}.get(config_id, 0)
    if num_real is None:
        num_real = 2 if hrtf else 0
    sc = oalgpu.Scene(api, sample_rate=48000, num_dry=4 if hrtf else 5, num_real=num_real,
                      num_sends=nsends, num_slots=nsends, wet_channels=4,
                      hrtf=hrtf, max_voices=nvoices, max_buffers=256, voices_per_group=vpg)
    sc.effects = []
    if config_id == 4:                  # EAX reverb, default preset, in every slot
        for slot in range(4):
            rev = oalgpu.Reverb(5)
            rev.update(oalgpu.ReverbProps.make(), 1.0)
            sc.set_slot_reverb(slot, rev)
            sc.effects.append(rev)
    if config_id == 5:                  # 65 536-tap exponentially decaying noise, BASELINE configs[4]
        lcg = synth.Lcg(0x5EED0005)
        ir = np.array([lcg.uniform(-1.0, 1.0) for _ in range(65536)], np.float32)
        ir *= np.exp(-np.arange(65536) / 12000.0).astype(np.float32) * 0.05
        conv = oalgpu.Convolution(4, ir)
        conv.set_target_gains([1.0, 0.0, 0.0, 0.0])
        sc.set_slot_convolution(0, conv)
        sc.effects.append(conv)
    if hrtf:
        # the post-process's ambisonic-to-binaural decoder as InitHrtfPanning builds it for a first-order HRTF device
        # (alc/panning.cpp:1100-1134): DirectHrtfState::build on the loaded data set, the cube layout, 400 Hz crossover
        sc.set_direct_hrtf_from_store(synth.AMBI_POINTS_1O, synth.AMBI_MATRIX_1O, synth.AMBI_ORDER_HF_GAIN_1O, 400.0)
    bufs = synth.scene_buffers(config_id, nvoices if voice_map is None else 256, sample_fmt)
    handles = [sc.add_buffer(b, oalgpu.FMT_SHORT if sample_fmt == "i16" else oalgpu.FMT_FLOAT) for b in bufs]
    script = synth.SceneScript(config_id, nvoices, voice_base, voice_map)
    for v in range(nvoices):
        sc.add_voice(handles[script.buffer_of(v, len(handles))], True, position=script.start_position(v))
    return sc, script


def param_array(oalgpu, script, voices, update):
    arr = (oalgpu.VoiceParams * len(voices))()
    for i, v in enumerate(voices):
        script.fill(arr[i], v, update)
    return arr


def _cpu_mix_worker(config_id, voice_base, nvoices, mhr_path, updates, target_seconds):
    """One replica of the reference CPU mixer on voices [voice_base, voice_base + nvoices) of the
    scene; returns (updates, seconds inside Voice::mix + MixDirectHrtf, oracle kind)."""
    import oracle_lib as ol
    from oalgpu import synth
    which = "ref" if ol.available("ref") else "port"
    L = ol.load(which)
    L.L.oal_set_simd(1)
    hrtf = config_id == 3
    if hrtf:
        L.hrtf_load(mhr_path)
    sc = ol.Scene(L, num_dry=4 if hrtf else 5, num_real=2 if hrtf else 0, hrtf=hrtf)
    if hrtf and L.kind == "reference":
        info = L.hrtf_raw()["info"]
        cc, hf, irsize = L.direct_hrtf_build(info.ir_size, False, synth.AMBI_POINTS_1O, synth.AMBI_MATRIX_1O, 4, 400.0,
                                             synth.AMBI_ORDER_HF_GAIN_1O)
        sc.set_direct_hrtf(cc, hf, 400.0 / info.sample_rate, irsize)
    bufs = synth.scene_buffers(config_id, max(nvoices, 256))
    handles = [sc.add_buffer(b, ol.FMT_FLOAT) for b in bufs]
    script = synth.SceneScript(config_id, nvoices, voice_base)
    for v in range(nvoices):
        sc.add_voice(handles[script.buffer_of(v, len(handles))], True, position=script.start_position(v))
        sc.set_params(v, script.fill(ol.VoiceParams(), v, 0))
    moving = [v for v in range(nvoices) if script.is_moving(v)]
    sc.mix(UPDATE_SAMPLES, post_process=hrtf)          # warm-up (not fading yet)
    if updates is None:
        t0 = time.perf_counter()
        sc.mix(UPDATE_SAMPLES, post_process=hrtf)
        one = time.perf_counter() - t0
        updates = int(max(3, min(200, target_seconds / max(one, 1e-6))))
    mix_time = 0.0
    for k in range(updates):
        for v in moving:                               # parameter side is not timed
            sc.set_params(v, script.fill(ol.VoiceParams(), v, k + 2))
        t0 = time.perf_counter()
        sc.mix(UPDATE_SAMPLES, post_process=hrtf)
        mix_time += time.perf_counter() - t0
    sc.close()
    return updates, mix_time, L.kind


def _cpu_replica(args):
    return _cpu_mix_worker(*args)


def host_cpu_info():
    """(physical cores, logical cpus, model name) of this host, from /proc/cpuinfo."""
    cores, logical, model = set(), 0, ""
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key == "processor":
                logical += 1
            elif key == "model name" and not model:
                model = val
            elif key == "physical id":
                phys = val
            elif key == "core id":
                core = val
                cores.add((phys, core))
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (logical or 1)
    usable = min(len(cores) or avail, avail)
    # a container may be held to fewer CPUs than it can see (cgroup v2 cpu.max / v1 cfs quota)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        usable = max(1, min(usable, int(quota)))
    return usable, avail, model


def cpu_baseline(config_id, nvoices, mhr_path, target_seconds=10.0):
    """The reference CPU mixer (compiled reference if it travelled, else the C restatement) on the same
    scene: (a) ONE thread -- the reference's real operating mode (one mixer thread per device); (b) P
    replicas, one per physical core of this host, the voices split evenly among them."""
    import multiprocessing as mp
    import oracle_lib as ol
    which = "ref" if ol.available("ref") else "port"
    if not ol.available(which):
        return None
    updates, mix_time, kind = _cpu_mix_worker(config_id, 0, nvoices, mhr_path, None, target_seconds)
    out = {"value": nvoices * updates / mix_time, "unit": "voices/s", "cores": 1,
           "kind": "reference" if kind == "reference" else "port",
           "sample": f"{nvoices} voices x {updates} updates of the same scene, 1 thread, "
                     f"{mix_time:.1f} s of Voice::mix + MixDirectHrtf"}
    cores, logical, model = host_cpu_info()
    out["cpu_model"] = model
    if cores > 1:
        per = (nvoices + cores - 1) // cores
        # every replica mixes its share of the voices for ~target_seconds / 2 of mixer time, side by side
        jobs = [(config_id, b, min(per, nvoices - b), mhr_path, None, target_seconds / 2)
                for b in range(0, nvoices, per)]
        try:
            ctx = mp.get_context("fork")
            t0 = time.perf_counter()
            with ctx.Pool(len(jobs)) as pool:
                res = pool.map(_cpu_replica, jobs)
            wall = time.perf_counter() - t0
            rate = sum(j[2] * r[0] / r[1] for j, r in zip(jobs, res))       # replicas run concurrently: rates add
            out["all_cores"] = {"value": rate, "unit": "voices/s", "cores": len(jobs),
                                "logical_cpus": logical, "cpu_model": model,
                                "sample": f"{len(jobs)} replicas (one per usable physical core) x {per} voices, "
                                          f"{min(r[0] for r in res)}-{max(r[0] for r in res)} updates each side by side, "
                                          f"{min(r[1] for r in res):.1f}-{max(r[1] for r in res):.1f} s inside the mixer "
                                          f"({wall:.1f} s wall with set-up)"}
        except Exception as e:                          # a sandbox without fork/semaphores: report the one-core leg only
            out["all_cores"] = {"error": repr(e)}
    return out


def mfma_block(kernel, voices, kernel_ms, moving):
    """What the matrix pipe executes per launch of the matrix-pipe HRTF voice kernel: per voice 2 ears x 5 tiles x 3
    K-chunks x 3 split-half products of v_mfma_f32_16x16x32_f16 (16 x 16 x 32 x 2 flop each), one more tile for a voice
    whose filter was replaced; against the dense f16 peak (MI355X_MICROARCH.md: 2.5 PFLOP/s)."""
    targs = [a.strip() for a in kernel[kernel.find("<") + 1:kernel.rfind(">")].split(",")] if "<" in kernel else []
    wave_mf = "VoiceWaveKernel<17, 64, 0" in kernel and len(targs) >= 5 and targs[4] == "true"     # (the fifth argument: MF)
    if not (wave_mf or kernel.startswith("VoiceWave16Kernel")):         # (voice_wave16.hip: the same tiles, a voice per wavefront)
        return None
    per = 16 * 16 * 32 * 2
    n = voices * 90 + moving * 18
    tf = n * per / (kernel_ms * 1e-3) / 1e12
    return {"instructions_per_launch": n, "executed_tflops": tf, "dense_f16_peak": 2500.0, "frac_of_dense_f16_peak": tf / 2500.0}


LDS_BYTES_PER_CLK_PER_CU = 256       # MI355X_MICROARCH.md, LDS: 64 dwords wide per clock (ds_read_b64 / b128 reach it)
NUM_CUS = 256
PEAK_CLOCK_HZ = 2.4e9                # MI355X peak engine clock


def lds_block(config_id, voices, kernel, kernel_ms):
    """The LDS side of the voice kernel, which is what its time follows: instructions, bytes and array-busy cycles
    per launch from the committed SQ counter passes of the same command (profiles/voice_kernel_sq_counters.json, written by
    tools/r6_evidence.sh on this tree's kernel sources), against the pipe's peak -- 256 B per clock and CU."""
    try:
        doc = json.load(open(os.path.join(ROOT, "profiles", "voice_kernel_sq_counters.json")))
        ent = doc["configs"][str(config_id)]
    except (OSError, ValueError, KeyError):
        return None
    if not _counters_current(doc):
        return {"refused": "profiles/voice_kernel_sq_counters.json was collected on other kernel sources (kernel_sources_sha256)"}
    if ent.get("voices") != voices or ent.get("kernel") != kernel:
        return None
    insts = ent["SQ_INSTS_LDS"]
    nbytes = insts * 64 * 8                 # a 64-lane ds_read_b64 / ds_write_b64 moves 512 B: the kernel's LDS traffic is 8-byte accesses
    peak = LDS_BYTES_PER_CLK_PER_CU * NUM_CUS * PEAK_CLOCK_HZ / 1e12
    tbs = nbytes / (kernel_ms * 1e-3) / 1e12
    cyc = kernel_ms * 1e-3 * PEAK_CLOCK_HZ
    return {"instructions_per_launch": insts, "bytes_per_launch": nbytes, "achieved": tbs, "peak": peak, "unit": "TB/s", "frac": tbs / peak,
            "array_busy_frac": ent["SQ_LDS_IDX_ACTIVE"] / NUM_CUS / cyc, "bank_conflict_cycles": ent.get("SQ_LDS_BANK_CONFLICT"),
            "wave_cycles_waiting_on_lds_frac": ent["SQ_WAIT_INST_LDS"] / ent["SQ_WAVE_CYCLES"],
            "valu_instructions_per_launch": ent.get("SQ_INSTS_VALU"),
            "note": "SQ_INSTS_LDS x 512 B (the kernels' LDS traffic is 8-byte accesses); array_busy = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel "
                    "cycles at 2.4 GHz): the pipe's busy share, not its byte rate, is the bound (DESIGN.md 3.4, 3.13)"}


def kernel_sources_hash():
    """sha256 over what decides the kernels' instruction streams -- openal-soft_amd/csrc/*.hip, *.hpp and the Makefile (flags), in
    name order.  Every committed counter file records the hash of the tree it was collected on (tools/r6_evidence.sh); a figure
    read from a file with another hash is refused (the line then carries null and says why)."""
    import glob, hashlib
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, "openal-soft_amd")
    for path in sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.hpp"))) + [os.path.join(pkg, "Makefile")]:
        h.update(os.path.basename(path).encode()); h.update(b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()


def _counters_current(doc):
    """a committed counter file belongs to this tree's kernels"""
    return doc.get("kernel_sources_sha256") == kernel_sources_hash()


def _file_source(name):
    """profiles/<name>@sha256:<first 16 hex digits>: which committed counter file a figure of the line was read from"""
    import hashlib
    path = os.path.join(ROOT, "profiles", name)
    try:
        return f"profiles/{name}@sha256:{hashlib.sha256(open(path, 'rb').read()).hexdigest()[:16]}"
    except OSError:
        return None


def _warm_code_pages():
    """On a fresh box the HIP runtime's and the library's code is paged in from disk as it is first executed; a path that runs
    for the first time inside a 1 ms timed block (a stream-wait the warm-up never needed, say) would put a disk read into it.
    Reading the mapped files once leaves their pages in the page cache, so that such a fault costs microseconds."""
    try:
        seen = set()
        for line in open("/proc/self/maps"):
            path = line.split(None, 5)[-1].strip() if line.count("/") else ""
            base = os.path.basename(path)
            if path in seen or not any(k in base for k in ("libamdhip64", "libhsa-runtime64", "liboalgpu", "libhsakmt", "libdrm")):
                continue
            seen.add(path)
            with open(path, "rb") as f:
                while f.read(1 << 22):
                    pass
    except OSError:
        pass


def join_ranks(oalgpu, sc, dist, torch, rank, world, local_rank, host_transport, tag):
    """the library's own exchange (oalgpu_comm_init / oalgpu_comm_init_host); torch.distributed only carries the 128-byte id"""
    if host_transport:
        sc.comm_init_host("/oalgpu_bench_%s_%s" % (os.environ.get("MASTER_PORT", "0"), tag), rank, world)
    else:
        idt = torch.zeros(128, dtype=torch.uint8, device=f"cuda:{local_rank}")
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(oalgpu.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
        sc.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)


def calibrate_shards(oalgpu, synth, api, args, V, rank, world, mhr, hrtf, post, dist, torch, local_rank, host_transport):
    """N > 1, before the voices are dealt: an equal-share scene (V voices per rank) through the library's sharded path -- every
    rank's own time per update (until ITS streams are idle) over 60 untimed updates behind 30 warm-up ones.  Rank 0 alone runs the
    effect slots and the post-process and receives the other ranks' bus blocks; its excess over the mean of the others is what
    the deal takes off its share (rank0_extra_us), the others' time per voice the conversion (us_per_voice)."""
    sc, script = build_scene(oalgpu, synth, api, args.config, V, rank * V, mhr, args.vpg, num_real=8 if args.config == 2 else None)
    if args.config == 2:
        dec_hf, dec_lf = synth.x71_decoder()
        sc.set_bformat_decoder(dec_hf, dec_lf)
    allv = list(range(V))
    moving = [v for v in allv if script.is_moving(v)]
    sc.set_params_batch(allv, param_array(oalgpu, script, allv, 0))
    blocks = [sc.param_block(moving, param_array(oalgpu, script, moving, k + 1)) for k in range(8)] if moving else []

    def run(n):
        for k in range(n):
            if blocks:
                sc.apply_block(blocks[k % len(blocks)])
            sc.mix(UPDATE_SAMPLES, post_process=post)
        sc.sync()
    tdev = "cpu" if host_transport else f"cuda:{local_rank}"

    def gathered(us):
        tt = torch.zeros(world, dtype=torch.float64, device=tdev)
        tt[rank] = us
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        return [float(x) for x in tt.cpu()]
    # the SAME scene on this rank alone, before the communicator is joined: what one GPU does with the workload every rank of the
    # job gets -- the figure a scaling efficiency divides by (the N = 1 line of this bench is the same config; this is the same
    # scene on the same machine in the same process)
    run(30)
    dist.barrier()
    t0 = time.perf_counter()
    run(60)
    solo = gathered((time.perf_counter() - t0) / 60 * 1e6)
    join_ranks(oalgpu, sc, dist, torch, rank, world, local_rank, host_transport, "cal")
    run(30)
    dist.barrier()
    t0 = time.perf_counter()
    run(60)
    per_rank = gathered((time.perf_counter() - t0) / 60 * 1e6)
    others = sum(per_rank[1:]) / max(world - 1, 1)
    sc.close()
    dist.barrier()
    return {"rank_us_per_update": per_rank, "rank0_extra_us": max(0.0, per_rank[0] - others), "us_per_voice": others / V,
            "solo_us_per_update": solo, "solo_voices": V,
            "note": "60 updates of an equal-share scene on this machine, every rank's own clock: first on every rank alone (solo_*: no "
                    "communicator, the rank's own post-process), then through the library's sharded path"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", type=int, default=None, choices=(2, 3, 4, 5),
                    help="BASELINE config (default: 3 = configs[2], the headline -- at every N, 4096 voices per GPU; 5 = configs[4], HRTF "
                         "voices + the 65536-tap convolution slot on rank 0, sharded over the GPUs)")
    ap.add_argument("--transport", default="rccl", choices=("rccl", "host"),
                    help="N > 1: how the ranks' bus blocks reach rank 0 -- the library's ncclReduce over xGMI, or its host-staged "
                         "transport (several processes on ONE GPU: a rehearsal of the N > 1 code path, not a scaling measurement)")
    ap.add_argument("--rank0-extra-us", type=float, default=None,
                    help="N > 1: what only rank 0 does per update (reduction of the ranks' blocks, effect slots, post-process), in us of "
                         "GPU time; its voice share shrinks by that much (default: per config, from profiles/)")
    ap.add_argument("--equal-shards", action="store_true", help="N > 1: deal every rank the same number of voices (A/B against the weighted deal)")
    ap.add_argument("--voices", type=int, default=None, help="voices per GPU (default 4096; 8192 for config 4)")
    ap.add_argument("--math", default="fast", choices=("fast", "exact"))
    ap.add_argument("--vpg", type=int, default=0, help="voices per workgroup (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mhr", default="default", choices=("default", "synth"),
                    help="HRTF data set: the reference's Default HRTF.mhr (tests/golden/default_hrtf.mhr) or the synthetic one")
    ap.add_argument("--repeats", type=int, default=5, help="extra K-step blocks timed after the contract's one (spread)")
    ap.add_argument("--fir", default="mfma", choices=("mfma", "valu"),
                    help="HRTF voices' dual-ear FIR: the matrix pipe in split half precision (the product default) or "
                         "packed fp32 VALU FMAs (OALGPU_CTX_FIR_VALU), for A/B runs")
    ap.add_argument("--preroll", type=int, default=None, help="untimed steps in front of the W warm-up steps, W included (default 2000)")
    ap.add_argument("--xflags", type=int, default=0, help="experiment bits or-ed into oalgpu_context_desc::flags")
    ap.add_argument("--resident", default="auto", choices=("auto", "on", "off"),
                    help="OALGPU_CTX_RESIDENT: one launch of the HRTF voice kernel stays on the machine over the updates of a block; every "
                         "step still is one oalgpu_param_block_apply + one oalgpu_mix_update with its own output (auto: on where the "
                         "library has the mode -- HRTF contexts without sends on one GPU, i.e. the headline config)")
    ap.add_argument("--static", action="store_true", help="experiment: no parameter block per step (no voice moves)")
    ap.add_argument("--run", type=int, default=0, metavar="B",
                    help="submit the steps B at a time through oalgpu_mix_update_run (one library call per B updates "
                         "instead of two per update); B must divide --steps and --warmup; 0 = one update per call")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher (WORLD_SIZE unset) would measure ONE GPU and call it N: it starts itself again as
    # the driver would -- one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1 -- and every rank then checks
    # that the world it finds is the one that was asked for.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush(); sys.stderr.flush()
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: the line would claim GPUs it did not run on "
                         f"(start it as python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}, or plain "
                         f"`python bench.py --gpus {args.gpus}`, which does that itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if args.config is None:
        # the SAME workload at every N (BASELINE configs[2], the one the metric is quoted on, 4096 voices per GPU: weak scaling) -- a
        # 1 -> 8 curve built from lines of different configs would show the configs' difference, not the scaling; configs[4]
        # (HRTF voices + the 65536-tap convolution slot on rank 0) is --config 5
        args.config = 3
    host_transport = world > 1 and args.transport == "host"
    if host_transport:
        local_rank = 0                       # every rank on the one GPU
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if host_transport:                   # (RCCL refuses two ranks on one device: the control plane goes over gloo)
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    import oalgpu
    from oalgpu import synth
    import oalmeasure

    if oalgpu.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU path exists)")
    torch.cuda.set_device(local_rank)
    # torch's lazy CUDA initialisation (hundreds of ms) happens HERE, not inside the first fence: there it left the GPU idle
    # right before the timed block, whose 20 steps then ran on ramping clocks (63 against 48.7 us per step at K = 20)
    torch.cuda.synchronize()
    # How the headline context runs its voice kernel (both are product modes, chosen by context flags):
    # resident -- ONE launch stays on its voices over the block, every step still one oalgpu_param_block_apply + one
    # oalgpu_mix_update with its own output -- pays for itself over a few dozen updates between two synchronisations (a launch's
    # first updates run at the launched pace and the block ends with the pipeline's drain: tools/resident_block_cost.py), so it is
    # what a K-step block of K >= 48 uses; shorter blocks launch per update, with the parameter block installed by the voice
    # kernel's own wavefronts (OALGPU_CTX_APPLY_IN_VOICE_KERNEL).  The library makes the same choice by itself for a host that
    # keeps its resident launches short (oalgpu_resident_set_short_run).
    hot = args.config == 3 and world == 1 and args.math == "fast" and args.fir == "mfma"
    want_resident = args.resident == "on" or (args.resident == "auto" and hot and args.steps >= 48)
    # (the parameter block installed by the voice kernel's own wavefronts pays on the headline config only: every VoiceWaveKernel has
    # the epilogue since round 5, and configs 2 / 4 / 5 measured 1.6 / 13 / 3.3 us per step SLOWER with it than with the parameter
    # kernel -- their steps hang on the post chain, which the deferred submission starts a library call later:
    # profiles/r5/apply_in_kernel_other_configs.txt; --xflags 16 selects it for an A/B)
    # (N > 1 runs the same scene sharded: the block is installed by each rank's voice kernel there too -- the library's condition for it
    # knows no communicator; only the resident launch is a single-GPU mode)
    hot_kernel = args.config == 3 and args.math == "fast" and args.fir == "mfma"
    mode_flags = (oalgpu.CTX_RESIDENT if want_resident else 0) | (oalgpu.CTX_APPLY_IN_VOICE_KERNEL if hot_kernel and args.resident != "off" else 0)
    api = oalgpu.Api(oalgpu.MATH_FAST if args.math == "fast" else oalgpu.MATH_EXACT, device=local_rank,
                     ctx_flags=(oalgpu.CTX_FIR_VALU if args.fir == "valu" else 0) | mode_flags | args.xflags)
    real_mhr = os.path.join(ROOT, "tests", "golden", "default_hrtf.mhr")
    use_real = args.mhr == "default" and os.path.exists(real_mhr)
    if use_real:
        with open(real_mhr, "rb") as f:
            mhr = f.read()
    else:
        mhr = synth.synth_mhr_bytes()
    api._mhr = mhr
    V = args.voices if args.voices else (8192 if args.config == 4 else 4096)
    hrtf = args.config in (3, 5)
    post = hrtf or args.config in (2, 4)     # effect slots / the speaker decode run with the post-process
    # config 2 is a 7.1 device: 5 ambisonic dry lines decoded to the 8 real output lines by the reference's
    # X71 decoder (BFormatDec, dual band) in the post-process; the host takes interleaved s16 PCM away
    # N > 1: the scene is world x V voices, dealt by cost class; rank 0 -- which alone sums the ranks' bus blocks,
    # runs the effect slots and the post-process, all on its post stream beside its own voice kernel -- gets fewer voices by what
    # that work takes (weighted_shards' rank0_extra), so that it is not the rank the others wait for.
    voice_map = None
    calibration = None
    shard_sizes = [V] * world
    if world > 1 and not args.equal_shards:
        from oalgpu.shard import voice_cost, weighted_shards
        probe = synth.SceneScript(args.config, V * world)
        nsends_of = (lambda v: v % 5) if args.config == 4 else (lambda v: 1 if args.config == 5 else 0)
        costs = [voice_cost(hrtf, 24, nsends_of(v), probe.filter_active(v)) for v in range(V * world)]
        # What only rank 0 does per update (it receives the ranks' blocks, runs the effect slots and the post-process) and what a
        # voice costs are MEASURED on this very machine before the deal: every rank runs 60 untimed updates of an equal-share
        # scene through the library's N > 1 path and reports its own time per update; rank 0's excess over the others is its
        # extra, the others' time over their voices the price of a voice (calibrate_shards below).  --rank0-extra-us overrides.
        calibration = calibrate_shards(oalgpu, synth, api, args, V, rank, world, mhr, hrtf, post, dist, torch, local_rank, host_transport)
        extra_us, voice_us = calibration["rank0_extra_us"], calibration["us_per_voice"]
        if args.rank0_extra_us is not None:
            extra_us = args.rank0_extra_us
        rank0_extra = extra_us / voice_us * (sum(costs) / len(costs))
        shards = weighted_shards(costs, world, rank0_extra=min(rank0_extra, 0.98 * sum(costs) / world * world / max(world - 1, 1)))
        shard_sizes = [len(sh) for sh in shards]
        voice_map = shards[rank]
        V = max(len(voice_map), 8)           # (a context holds at least one workgroup of voices)
        if len(voice_map) < V:               # rank 0 with (almost) nothing to mix: pad with voices of its own that stay silent
            voice_map = voice_map + [voice_map[-1] if voice_map else 0] * (V - len(voice_map))
    sc, script = build_scene(oalgpu, synth, api, args.config, V, rank * V, mhr, args.vpg,
                             num_real=8 if args.config == 2 else None, voice_map=voice_map)
    if args.config == 2:
        dec_hf, dec_lf = synth.x71_decoder()
        sc.set_bformat_decoder(dec_hf, dec_lf)
        sc.set_output(oalgpu.OUT_I16, 0.0, 22222)

    all_voices = list(range(V))
    moving = [v for v in all_voices if script.is_moving(v)]
    sc.set_params_batch(all_voices, param_array(oalgpu, script, all_voices, 0))
    # every step applies a parameter block (new directions for the moving quarter of the voices);
    # the blocks are resident in HBM and a ring of 96 different ones is cycled through, so that a
    # long run needs neither gigabytes of records nor minutes of host-side preparation
    total_steps = args.warmup + 2 * args.steps
    nblocks = min(total_steps, 96)
    blocks = [sc.param_block(moving, param_array(oalgpu, script, moving, k + 1)) for k in range(nblocks)]

    # N > 1: the library's own multi-GPU path (RCCL inside liboalgpu.so: oalgpu_comm_init, then every
    # oalgpu_mix_update issues the ncclReduce of the bus block on its post stream and only rank 0 runs
    # the effects and the post-process) -- what a C++ host would call.  torch.distributed only carries
    # the 128-byte id to the other ranks and the contract's barrier / max-over-ranks.
    force_sharded = os.environ.get("OALGPU_FORCE_SHARDED") == "1"      # exercise the N>1 path on one GPU
    if world > 1 or force_sharded:
        if dist is None:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        join_ranks(oalgpu, sc, dist, torch, rank, world, local_rank, host_transport, "run")
        comm_seen = sc.comm_info()
        if comm_seen[1] != world or (comm_seen[2] not in (-1, world)):
            raise SystemExit(f"bench.py: the library's exchange sees {comm_seen} but the job has {world} ranks")

    comm_seen = locals().get("comm_seen")
    B = args.run
    if B and (args.steps % B or args.warmup % B):
        raise SystemExit("--run B: B must divide --steps and --warmup")

    def step(k):
        if B:                        # B updates per library call
            if k % B == 0:
                sc.mix_run([blocks[(k + j) % nblocks] for j in range(B)], UPDATE_SAMPLES, post)
            return
        if not args.static:
            sc.apply_block(blocks[k % nblocks])
        sc.mix(UPDATE_SAMPLES, post_process=post)

    def fence():
        if dist is not None:
            dist.barrier()
        sc.sync()
        torch.cuda.synchronize()

    # Pre-roll: the scene is brought to its steady state (every voice mid-buffer, the two-stream
    # pipeline full, GPU clocks up) before the W warm-up and the K timed steps the contract names;
    # without it a short run (K = 20 is 1 ms) measures the clock ramp, which takes tens of milliseconds:
    # the K = 20 block after 5 / 50 / 150 / 400 / 1500 untimed steps ran at 59 / 54 / 51 / 51 / 48 us per step
    # (two runs each, profiles/r3/preroll_sweep.txt).
    preroll = max(0, (2000 if args.preroll is None else args.preroll) - args.warmup)
    if B:
        preroll -= preroll % B
    # The step loop gains only ~10 us on the GPU per step (host 32 us, GPU 42.5 us), so a block that starts from an empty
    # queue -- as the contract's does -- has no lead to absorb a hiccup of the submitting thread.  Two sources were found:
    # (1) the interpreter's garbage collector (the scene's parameter arrays are ~10^5 ctypes objects to walk; nothing in
    # the loop creates cycles: collect once, then keep it off); (2) the HIP runtime, which reclaims the commands of
    # everything submitted since the last device-wide synchronisation inside one of the first calls AFTER the next one
    # -- behind 2000 unsynchronised pre-roll steps that was a single 150-270 us call within the first ten steps of the
    # timed block (63 / 52 / 57 / 71 us per step at K = 20 on some runs, 47 on others; the repeat blocks, 20 steps
    # behind their fence, never showed it).  The pre-roll therefore synchronises every 25 steps: K = 20 blocks then
    # measure 45.2-46.7 us per step in six of six runs (profiles/r3/contract_block.txt).
    _warm_code_pages()
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    tdev = "cpu" if host_transport else f"cuda:{local_rank}"

    def timed_block(first_step):
        """K steps bracketed as the contract says; returns (seconds, max over ranks; this rank's own seconds until ITS streams
        were idle, before the closing barrier)."""
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(first_step + k)
        if dist is not None:             # (one rank: the closing fence below is this rank's own clock as well)
            sc.sync()
        own = time.perf_counter() - t0
        fence()
        e = time.perf_counter() - t0
        if dist is None:
            own = e
        if dist is not None:
            tt = torch.tensor([e], dtype=torch.float64, device=tdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e = float(tt.item())
        return e, own

    # the K-step block behind ONLY the W warm-up steps the command line names (clocks still ramping, the runtime's queues cold):
    # reported beside the pre-rolled block as config.cold_block_ms_per_step, so that what the pre-roll is worth is in the record
    for k in range(args.warmup):
        step(k)
    fence()
    cold_elapsed, _ = timed_block(args.warmup)
    sync_every = max(25, args.steps) if want_resident else 25      # (a resident launch lives from one synchronisation to the next)
    for k in range(preroll):
        step(k)
        if k % sync_every == sync_every - 1:
            fence()
    for k in range(args.warmup):
        step(k)
    fence()
    elapsed, own_elapsed = timed_block(args.warmup)
    rank_ms = [own_elapsed / args.steps * 1e3]
    if dist is not None:
        tt = torch.zeros(world, dtype=torch.float64, device=tdev)
        tt[rank] = own_elapsed / args.steps * 1e3
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        rank_ms = [float(x) for x in tt.cpu()]

    # ---- spread: the same K-step block a few more times (each bracketed like the contract's one)
    extra = []
    for r in range(max(0, args.repeats)):
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(args.warmup + (r + 1) * args.steps + k)
        fence()
        e = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([e], dtype=torch.float64, device=tdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e = float(tt.item())
        extra.append(e / args.steps * 1e3)
    # the resident launch's own time (HIP events bound to its dispatch: they cost the launch call ~15 us of host time, so they
    # are switched on only here, behind the timed blocks): the same K-step block a few more times
    res0 = res1 = None
    if want_resident:
        sc.resident_set_timing(True)
        res0 = sc.resident_stats()
        for r in range(max(1, min(args.repeats, 3))):
            for k in range(args.steps):
                step(args.warmup + (r + 7) * args.steps + k)
            fence()
        res1 = sc.resident_stats()
        sc.resident_set_timing(False)

    # ---- end-to-end latency of ONE update through the boundary as a host uses it: the moving voices'
    # oalgpu_voice_params records go in from host memory (biquad design + H2D inside
    # oalgpu_voice_set_params), the update runs, the output lines come back (D2H + sync); nothing
    # overlaps.  Reported beside the throughput figure, never as `value`.
    e2e_ms = e2e_p90_ms = None
    if world == 1 and moving:
        recs = [param_array(oalgpu, script, moving, 500 + k) for k in range(8)]
        take = (lambda: sc.read_output(UPDATE_SAMPLES, 8)) if args.config == 2 else sc.dry
        for k in range(3):
            sc.set_params_batch(moving, recs[k]); sc.mix(UPDATE_SAMPLES, post_process=post); take()
        n_e2e = 40
        each = []
        for k in range(n_e2e):
            t0 = time.perf_counter()
            sc.set_params_batch(moving, recs[k % len(recs)])
            sc.mix(UPDATE_SAMPLES, post_process=post)
            take()       # oalgpu_read_dry (dry + real lines) / oalgpu_read_output (8-channel s16 PCM): D2H, synchronises
            each.append((time.perf_counter() - t0) * 1e3)
        each.sort()
        e2e_ms = each[len(each) // 2]         # the median of 40: a host that is pre-empted once does not decide the figure
        e2e_p90_ms = each[(len(each) * 9) // 10]

    # ---- end-to-end THROUGHPUT through the pipelined boundary: per update the moving voices' 24-byte move records go
    # in from host memory as they are (oalgpu_voice_move_async: one copy into a ring slot -- device memory behind the BAR
    # where the box has a large one, pinned host memory otherwise -- and the installing kernel, in front of the next voice
    # kernel, evaluates getCoeffs' index half itself), the update runs with its post-process, the stereo output comes back
    # (the post-process kernel stores it into the host's pinned ring slot; collected two updates late).  Nothing waits for
    # anything but the output of two updates ago.
    e2e_tput = None
    if world == 1 and moving and hrtf and not B:
        def moves_of(update):
            m = np.zeros(len(moving), oalgpu.MOVE_DTYPE)
            for i, v in enumerate(moving):
                ev, az, gain = script.direction(v, update)
                m[i] = (v, ev, az, 2.0, 0.0, gain)
            return m
        mv = [moves_of(700 + k) for k in range(8)]
        outbuf = np.empty((2, 1024), np.float32)

        def run(n):
            tickets = []
            busy = 0.0
            sc.sync()
            t0 = time.perf_counter()
            for k in range(n):
                h0 = time.perf_counter()
                sc.move_async(mv[k % len(mv)])
                sc.mix(UPDATE_SAMPLES, post_process=post)
                tickets.append(sc.read_output_async())
                busy += time.perf_counter() - h0
                if k >= 2:
                    sc.output_wait(tickets[k - 2], outbuf)
            for t in tickets[-2:]:
                sc.output_wait(t, outbuf)
            return time.perf_counter() - t0, busy
        run(50)
        n_tp = 400
        each = sorted(run(n_tp) for _ in range(3))
        dt, busy = each[1]
        oalmeasure.pipelined_run(sc, mv, 50, UPDATE_SAMPLES, post)
        native = sorted(oalmeasure.pipelined_run(sc, mv, n_tp, UPDATE_SAMPLES, post) for _ in range(3))[1]
        submit_s = sorted(oalmeasure.submit_cost(sc, mv, 200, UPDATE_SAMPLES, post) for _ in range(3))[1]
        e2e_tput = {"e2e_voices_per_s": V * n_tp / dt, "ms_per_update": dt / n_tp * 1e3, "host_submit_share": busy / dt,
                    "native_loop": {"e2e_voices_per_s": V * n_tp / native[0], "ms_per_update": native[0] / n_tp * 1e3,
                                    "host_submit_share": native[1] / native[0],
                                    "note": "the same loop written in C++ (tools/measure: oalmeasure_pipelined_run): no python / ctypes per call",
                                    "submit_us_unqueued": submit_s * 1e6, "host_share_unqueued": submit_s / (native[0] / n_tp),
                                    "note_unqueued": "the three submitting calls timed with every update's output waited for before the next "
                                                     "is submitted (oalmeasure_submit_cost): host_submit_share above also counts the time a call "
                                                     "spends blocked inside the runtime behind the queue the GPU is draining"},
                    "updates": n_tp, "moved_voices_per_update": len(moving),
                    "note": "median of 3 runs; per update: oalgpu_voice_move_async (raw 24-byte records into a ring slot, "
                            "ApplyMovesKernel evaluates the HRIR-blend indices) + oalgpu_mix_update + post-process + "
                            "oalgpu_read_output_async (the post-process kernel fills the host's slot), output collected "
                            "two updates late (oalgpu_output_wait); host_submit_share = the calling thread's time inside those "
                            "three calls / wall time (python + ctypes included)"}

    # ---- instrumented pass: HIP events on the context's stream around each voice-kernel launch
    sc.set_timing(True)
    vk = []
    tot = []
    for k in range(args.steps):
        sc.apply_block(blocks[(args.warmup + args.steps + k) % nblocks])
        sc.mix_voices(UPDATE_SAMPLES)
        a, b = sc.last_update_ms()
        tot.append(a)
        vk.append(b)
        if post and rank == 0:
            sc.post_process(UPDATE_SAMPLES)
    sc.sync()
    sc.set_timing(False)
    vk_ms = float(np.mean(vk))
    # The resident launch (OALGPU_CTX_RESIDENT): the voice kernel of the timed blocks is ONE launch per block; its own duration
    # (HIP events bound to the dispatch, collected by the library when the launch has ended) over the updates it covered is
    # the kernel's time per update -- cold start and any wait for the host's doorbells included.  The launched kernel's time of
    # the pass above stays in the line as kernel_ms_launched.
    resident = None
    vk_launched_ms = vk_ms
    if want_resident and res0 and res1 and res1["timed_updates"] > res0["timed_updates"] and not res1["failed"]:
        upd = res1["timed_updates"] - res0["timed_updates"]
        lau = res1["timed_launches"] - res0["timed_launches"]
        kms = res1["timed_kernel_ms"] - res0["timed_kernel_ms"]
        vk_ms = kms / upd
        resident = {"launches": lau, "updates": upd, "launch_ms_mean": kms / max(lau, 1), "updates_per_launch": upd / max(lau, 1),
                    "kernel_ms_per_update": vk_ms, "door_in_device_memory": bool(res1["door_in_device_memory"]),
                    "parks_total": res1["parks"], "launches_total": res1["launches"],
                    "waits_us_per_update": {k[:-3]: (res1[k] - res0[k]) / upd for k in res1 if k.endswith("_us")},
                    "note": "K-step blocks like the timed ones, run behind them with the launches' events on: one launch of the voice kernel "
                            "per block, ended by the block's closing oalgpu_sync; launch_ms_mean is what rocprofv3's kernel trace shows "
                            "per call of VoiceWaveKernel<..., true> for such blocks"}
    # the same clock around an EMPTY kernel: what the dispatch-bound events include besides a kernel's own run time
    # (rocprofv3's kernel trace reports the voice kernel about this much shorter, profiles/README.md)
    event_floor_ms = oalmeasure.event_floor_ms(sc, 200) if sc.voice_kernel_name().startswith(("VoiceWaveKernel", "VoiceWave16Kernel", "VoiceRowsKernel")) else None

    if rank == 0:
        nvoices_total = sum(shard_sizes)
        bytes_per_launch = BYTES_PER_VOICE_UPDATE[args.config] * V
        flops_per_launch = FLOPS_PER_VOICE_UPDATE[args.config] * V
        hbm_achieved = bytes_per_launch / (vk_ms * 1e-3) / 1e9
        achieved = flops_per_launch / (vk_ms * 1e-3) / 1e12
        # HBM bytes per launch of the voice kernel from the committed rocprofv3 PMC passes
        # (profiles/voice_kernel_traffic.json: FETCH_SIZE / WRITE_SIZE collected and corrected
        # as MI355X_MICROARCH.md prescribes); None when the profile was taken on another config
        traffic = None
        traffic_source = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "voice_kernel_traffic.json")))
            if not _counters_current(tj):
                traffic_source = "refused: profiles/voice_kernel_traffic.json was collected on other kernel sources (kernel_sources_sha256)"
            for ent in (tj.get("configs", {}).values() if _counters_current(tj) else ()):       # (config 4 has two entries: rows in LDS, stream rows)
                if ent.get("config") == args.config and ent.get("voices") == V and ent.get("kernel") == sc.voice_kernel_name():
                    traffic = ent["hbm_bytes_per_launch"]
                    traffic_source = _file_source("voice_kernel_traffic.json")
        except (OSError, ValueError, KeyError):
            pass
        out = {
            # BASELINE.json's metric string verbatim for the two HRTF-stereo configs -- configs[2], the headline, and
            # configs[4], the same voices with a send into the convolution slot, which `--gpus N` runs (config.workload says
            # which): the "fraction of HBM roofline" half is roofline.hbm_frac, `value` the voices/s half
            "metric": {3: "mixed voices/sec @48kHz 1024-sample update, HRTF stereo; fraction of HBM roofline",
                       2: "mixed voices/sec @48kHz 1024-sample update, bsinc24 -> 7.1 dry bus",
                       4: "mixed voices/sec @48kHz 1024-sample update, 7.1 dry bus + 4 EAX reverb slots",
                       5: "mixed voices/sec @48kHz 1024-sample update, HRTF stereo; fraction of HBM roofline"}[args.config],
            "value": nvoices_total * args.steps / elapsed,
            "unit": "voices/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "hbm_roofline_frac": hbm_achieved / HBM_PEAK_GBS,     # the second half of the metric string
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{world} ranks, each: " if world > 1 else "") + f"BASELINE configs[{args.config - 1}]: {nvoices_total // world} mono f32 voices per GPU "
                                   f"(44.1k->48k, bsinc24"
                                   + ((", HRTF Default HRTF.mhr (irSize 64), " if use_real else
                                       ", HRTF synthetic .mhr with Default-HRTF geometry irSize 64, ")
                                      + "dual-ear FIR + MixDirectHrtf" if hrtf else
                                      (", 5-line dry mix + X71 dual-band decode to 8 speaker lines" if args.config == 2 else ", 5-line dry mix"))
                                   + {4: ", v%5 sends into 4 reverb slots", 5: ", one send into a 65536-tap convolution slot"}.get(args.config, "")
                                   + "), 25% filtered, every 4th voice moving",
                       "voices_total": nvoices_total, "update_samples": UPDATE_SAMPLES,
                       "voice_kernel_mode": ("resident launch (OALGPU_CTX_RESIDENT): one launch per K-step block" if want_resident else
                                             ("launch per update, parameter block installed by the voice kernel (OALGPU_CTX_APPLY_IN_VOICE_KERNEL)"
                                              if mode_flags & oalgpu.CTX_APPLY_IN_VOICE_KERNEL else "launch per update")),
                       "preroll_steps": preroll, "cold_block_ms_per_step": cold_elapsed / args.steps * 1e3, "math_mode": args.math,
                       "voices_per_rank": shard_sizes, "rank_ms_per_step": rank_ms,
                       "transport": (args.transport if world > 1 else None),
                       "rccl_ranks": (comm_seen[2] if comm_seen and comm_seen[3] == "rccl" else None),
                       "comm": ({"rank": comm_seen[0], "world": comm_seen[1], "transport_ranks": comm_seen[2], "kind": comm_seen[3]} if comm_seen else None),
                       "calibration": calibration,
                       # N > 1: rank 0's own rate on the equal-share scene of THIS config before the communicator was joined, and the job's
                       # rate over N times that (what the driver's curve divides by is the N = 1 line; this is the same figure measured
                       # in this very run)
                       "single_gpu_same_workload": ({"voices_per_s": calibration["solo_voices"] / (calibration["solo_us_per_update"][0] * 1e-6),
                                                     "ms_per_step": calibration["solo_us_per_update"][0] * 1e-3, "voices": calibration["solo_voices"],
                                                     "config": args.config} if calibration and "solo_us_per_update" in calibration else None),
                       "scaling_efficiency": ((nvoices_total * args.steps / elapsed)
                                              / (world * calibration["solo_voices"] / (calibration["solo_us_per_update"][0] * 1e-6))
                                              if calibration and "solo_us_per_update" in calibration else None),
                       "realtime_voices": nvoices_total * args.steps / elapsed / 46.875,
                       "e2e_ms_per_update": e2e_ms, "e2e_ms_per_update_p90": e2e_p90_ms if e2e_ms is not None else None,
                       "e2e_throughput": e2e_tput,
                       "e2e_note": "one update alone through the C-ABI from host memory: oalgpu_voice_set_params of the "
                                   f"{len(moving)} moving voices (host biquad design + H2D) + oalgpu_mix_update + "
                                   "oalgpu_read_dry (D2H, sync); no overlap -- a latency, not the throughput `value` is",
                       "repeat_ms_per_step": {"n": len(extra), "median": float(np.median(extra)) if extra else None,
                                              "min": min(extra) if extra else None, "max": max(extra) if extra else None},
                       "parallelism": f"voice-shard x{world}" + ((" + the bus block summed to rank 0 through host-staged shared memory (a rehearsal of the code path on one device, not RCCL)"
                                                                       if host_transport else " + ncclReduce of the bus block to rank 0, issued by the library") if world > 1 else "")},
            # 68 flop/B: the path sits above the fp32 ridge (157.3 TFLOP/s / 8 TB/s = 20 flop/B), so the binding
            # roofline is arithmetic.  `achieved` prices the ALGORITHMIC fp32 flops of the path against
            # the fp32 peak, 157.3 TFLOP/s -- the same figure for v_pk_fma_f32 and for fp32 MFMA on gfx950 -- as
            # rounds 1 and 2 did.  The HRTF kernel's FIR (73 % of those flops) runs on the matrix pipe in split half
            # precision (three v_mfma_f32_16x16x32_f16 products per fp32 product, Toeplitz tiles: DESIGN.md 3.1);
            # what the pipe EXECUTES for it is under `mfma`.  `hbm_frac` is the "fraction of HBM roofline"
            # BASELINE's metric string names.  kernel_ms: HIP events bound to the dispatch (hipExtLaunchKernel).
            # `bound`: what the counters name.  Neither HBM (hbm_frac) nor the matrix pipe (mfma) is near its limit; the kernel's
            # time follows its LDS instruction stream at two wavefronts per SIMD (`lds`): that is the binding unit.  `achieved` /
            # `peak` / `frac` stay the algorithmic fp32 figures of rounds 1-3, for continuity.
            "roofline": {"bound": "lds", "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_PEAK_TFLOPS, "traffic": traffic,
                         # (traffic, lds and mfma.instructions are NOT measured in this run: they are the committed rocprofv3 PMC passes
                         # of the same command on the builder's box, per launch of the launched kernel -- the files and their hashes)
                         "traffic_source": traffic_source, "counters_source": _file_source("voice_kernel_sq_counters.json"),
                         "kernel": sc.voice_kernel_name() + (" [resident launch]" if resident else ""), "kernel_ms": vk_ms,
                         "kernel_ms_launched": vk_launched_ms, "resident": resident, "event_floor_ms": event_floor_ms,
                         "lds": lds_block(args.config, V, sc.voice_kernel_name(), vk_ms),
                         "mfma": mfma_block(sc.voice_kernel_name(), V, vk_ms, len(moving)),
                         "flops_per_launch": flops_per_launch, "bytes_per_launch": bytes_per_launch,
                         "hbm_achieved": hbm_achieved, "hbm_peak": HBM_PEAK_GBS, "hbm_unit": "GB/s",
                         "hbm_frac": hbm_achieved / HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline and args.config in (2, 3):
            if hrtf and not use_real:
                tmp_mhr = os.path.join(tempfile.gettempdir(), f"oalgpu_bench_{os.getpid()}.mhr")
                synth.write_synth_mhr(tmp_mhr)
            cb = cpu_baseline(args.config, V, real_mhr if use_real else (tmp_mhr if hrtf else ""))
            if cb:
                out["cpu_baseline"] = cb
        # anything native code left in C stdio buffers (the RCCL banner) goes out before the line
        C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    sc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
