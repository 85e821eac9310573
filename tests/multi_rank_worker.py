"""Worker of tests/test_multi_rank.py: one rank of a voice-sharded HRTF scene on the CPU oracle,
driven by the SAME orchestration bench.py uses on GPUs (oalgpu.shard.ShardedMixer), over gloo.
Rank 0 also mixes the whole scene unsharded and compares.  Exit code 0 = parity."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd"))
import oracle_lib as ol                      # noqa: E402
from oalgpu import synth                     # noqa: E402  (host-side helpers only; no GPU call)
from oalgpu.shard import ShardedMixer, shard_range   # noqa: E402

N_VOICES, N_UPDATES, N_DRY = 22, 4, 4
SIZES = [1024, 1024, 700, 1024]
RTOL, ATOL = 2e-5, 1e-7


class OracleEngine:
    """ShardedMixer engine on the CPU oracle: the bus block is [dry+real lines | HrtfAccumData]."""

    def __init__(self, L, scene, cc, hf, xover, irsize):
        self.L, self.sc = L, scene
        self.bus = torch.zeros((N_DRY + 2) * 1024 + 1152 * 2, dtype=torch.float32)
        self.carry_on = False
        self.carry = np.zeros((1152, 2), np.float32)          # HrtfAccumData tail, carrying rank only
        self.splitters = []
        for _ in range(N_DRY):
            s = ol.Splitter()
            L.L.oal_splitter_init(C.byref(s), xover)
            self.splitters.append(s)
        self.cc, self.hf, self.irsize = cc, hf, irsize
        self.accum_ptr = C.cast(L.L.oal_scene_hrtf_accum(scene.h), C.c_void_p)

    def set_carry(self, on):
        self.carry_on = on

    def mix_voices(self, n):
        C.memset(self.accum_ptr, 0, 1152 * 2 * 4)             # partial accumulator of THIS update
        self.sc.mix(n, post_process=False)
        b = self.bus.numpy()
        b[:(N_DRY + 2) * 1024] = self.sc.dry().ravel()
        acc = self.sc.hrtf_accum()
        if self.carry_on:
            acc = acc + self.carry                            # the reduction adds the carried tail once
        b[(N_DRY + 2) * 1024:] = acc.ravel()

    def bus_tensor(self):
        return self.bus

    def collective(self):
        import contextlib
        return contextlib.nullcontext()

    def post_process(self, n, run=True):
        if not run:
            return
        b = self.bus.numpy()
        lines = b[:(N_DRY + 2) * 1024].reshape(N_DRY + 2, 1024)
        acc = b[(N_DRY + 2) * 1024:].reshape(1152, 2).copy()
        left, right = lines[N_DRY].copy(), lines[N_DRY + 1].copy()
        self.splitters = self.L.mix_direct_hrtf(left, right, lines[:N_DRY].copy(), acc, self.splitters, self.hf,
                                                self.cc, self.irsize, n)
        lines[N_DRY], lines[N_DRY + 1] = left, right
        self.carry = acc                                      # shifted by MixDirectHrtf: next update's tail
        b[(N_DRY + 2) * 1024:] = acc.ravel()


def build(L, script_all, lo, hi, bufs, cc, hf, xover):
    sc = L.make_scene(num_dry=N_DRY, num_real=2, hrtf=True)
    sc.set_direct_hrtf(cc, hf, xover, 64)
    handles = [sc.add_buffer(b, ol.FMT_FLOAT) for b in bufs]
    for gv in range(lo, hi):
        sc.add_voice(handles[gv % len(handles)], True, position=(gv * 7919) % 4000)
        sc.set_params(gv - lo, script_all.fill(ol.VoiceParams(), gv, 0))
    return sc


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = ol.load("port")
    L.L.oal_set_simd(1)
    L.hrtf_load(os.environ["OAL_TEST_MHR"])
    rng = np.random.default_rng(99)
    bufs = [rng.uniform(-1, 1, 5000).astype(np.float32) for _ in range(3)]
    cc = np.zeros((N_DRY, 128, 2), np.float32)
    cc[:, :64] = rng.uniform(-0.2, 0.2, (N_DRY, 64, 2))
    hf, xover = [1.0, 0.8, 0.8, 0.8], 400.0 / 48000.0
    script = synth.SceneScript(3, N_VOICES)
    lo, hi = shard_range(N_VOICES, rank, world)
    sc = build(L, script, lo, hi, bufs, cc, hf, xover)
    engine = OracleEngine(L, sc, cc, hf, xover, 64)
    mixer = ShardedMixer(engine, dist, rank, world)
    whole = build(L, script, 0, N_VOICES, bufs, cc, hf, xover) if rank == 0 else None
    ok = True
    for k in range(N_UPDATES):
        n = SIZES[k]
        if k > 0:
            for gv in range(lo, hi):
                if script.is_moving(gv):
                    sc.set_params(gv - lo, script.fill(ol.VoiceParams(), gv, k))
        mixer.update(n)
        if rank == 0:
            if k > 0:
                for gv in range(N_VOICES):
                    if script.is_moving(gv):
                        whole.set_params(gv, script.fill(ol.VoiceParams(), gv, k))
            whole.mix(n, post_process=True)
            want = np.concatenate([whole.dry().ravel(), whole.hrtf_accum().ravel()]).astype(np.float64)
            got = engine.bus_tensor().numpy().astype(np.float64)
            err = np.max(np.abs(got - want))
            bound = RTOL * np.max(np.abs(want)) + ATOL
            print(f"update {k}: max err {err:.3e} (bound {bound:.3e}, max|ref| {np.max(np.abs(want)):.3e})", flush=True)
            ok = ok and err <= bound and np.max(np.abs(want)) > 1e-4
    # integer voice state of every shard equals the unsharded scene's (exchange through rank 0)
    mine = [sc.voice_state(v) for v in range(hi - lo)]
    ints = torch.tensor([[s.play_state, s.position, s.position_frac, s.has_buffer, s.fading] for s in mine],
                        dtype=torch.int64)
    gathered = [torch.zeros((shard_range(N_VOICES, r, world)[1] - shard_range(N_VOICES, r, world)[0], 5),
                            dtype=torch.int64) for r in range(world)] if rank == 0 else None
    if world > 1:
        # gloo gather needs equal shapes: pad to the largest shard
        m = max(shard_range(N_VOICES, r, world)[1] - shard_range(N_VOICES, r, world)[0] for r in range(world))
        pad = torch.zeros((m, 5), dtype=torch.int64)
        pad[:ints.shape[0]] = ints
        out = [torch.zeros((m, 5), dtype=torch.int64) for _ in range(world)] if rank == 0 else None
        dist.gather(pad, out, dst=0)
        if rank == 0:
            gathered = [out[r][:shard_range(N_VOICES, r, world)[1] - shard_range(N_VOICES, r, world)[0]] for r in range(world)]
    else:
        gathered = [ints]
    if rank == 0:
        allints = torch.cat(gathered).tolist()
        for gv in range(N_VOICES):
            s = whole.voice_state(gv)
            if allints[gv] != [s.play_state, s.position, s.position_frac, s.has_buffer, s.fading]:
                print("voice state mismatch", gv, allints[gv], flush=True)
                ok = False
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, src=0)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
