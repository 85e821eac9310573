"""Voice sharding across the GPUs of one node: voices are independent given
their parameters, so every rank owns a contiguous shard of them and mixes it into PARTIAL buses;
the only exchange is one sum-reduce of the bus block [dry+real lines | wet buses | HrtfAccumData]
to rank 0 per update (RCCL over xGMI on the GPU box, ~41 KB), after which rank 0 -- the only rank
that carries the HRTF accumulator tail between updates -- runs the effects and the HRTF
post-process (alc/alu.cpp:2209-2257, :289-298) on the summed buses.

The orchestration is backend-agnostic: ``engine`` is anything with
    set_carry(bool) / mix_voices(n) / bus_tensor() -> torch tensor aliasing the bus block /
    collective() -> context manager the reduce is issued under / post_process(n, run)
bench.py passes the HIP context (OverlappedGpuEngine, nccl); tests/test_multi_rank.py passes the
CPU oracle (gloo, world_size 2) to check the scheme itself against an unsharded scene."""
import contextlib


def shard_range(total_voices, rank, world):
    """Contiguous, near-equal shards: rank r owns [lo, hi)."""
    base, extra = divmod(total_voices, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedMixer:
    def __init__(self, engine, dist=None, rank=0, world=1):
        self.engine, self.dist, self.rank, self.world = engine, dist, rank, world
        engine.set_carry(rank == 0)

    def update(self, samples_to_do):
        e = self.engine
        e.mix_voices(samples_to_do)
        if self.dist is not None and (self.world > 1 or getattr(e, "always_reduce", False)):
            with e.collective():
                self.dist.reduce(e.bus_tensor(), dst=0, op=self.dist.ReduceOp.SUM)
        e.post_process(samples_to_do, self.rank == 0)


class GpuEngine:
    """The HIP context behind the ShardedMixer interface.  ``torch_stream`` is the stream RCCL is
    ordered on; the context runs on it too (oalgpu_set_stream), so no extra synchronisation."""

    def __init__(self, scene, torch, device_index, torch_stream):
        self.sc = scene
        scene.set_stream(torch_stream.cuda_stream)
        ptr, nfloats, _ = scene.bus_device_ptr()

        class _Bus:
            __cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 2}
        self._bus = torch.as_tensor(_Bus(), device=f"cuda:{device_index}")

    def set_carry(self, on):
        self.sc.set_carry_accum(on)

    def mix_voices(self, n):
        self.sc.mix_voices(n)

    def bus_tensor(self):
        return self._bus

    def collective(self):
        return contextlib.nullcontext()

    def post_process(self, n, run):
        if run:
            self.sc.post_process(n)


class OverlappedGpuEngine:
    """The HIP context on its OWN two streams (FAST HRTF contexts): the voice kernel of update
    k+1 runs on the main stream while the partial-bus reduction, the RCCL reduce and the
    post-process of update k run on the context's post stream, which torch sees as an
    ExternalStream -- the collective is issued with that stream current, so torch.distributed
    orders it between the reduction before and the post-process after it."""

    def __init__(self, scene, torch, device_index):
        self.sc = scene
        self.torch = torch
        ptr, nfloats, _ = scene.bus_device_ptr()

        class _Bus:
            __cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 2}
        self._bus = torch.as_tensor(_Bus(), device=f"cuda:{device_index}")
        self._post = torch.cuda.ExternalStream(scene.post_stream(), device=f"cuda:{device_index}")
        # the post stream is made torch's current stream ONCE (nothing else of torch runs in the update
        # loop, and the library never looks at torch's current stream): entering a stream context per
        # update costs ~5 us of host time in a loop whose host side is as long as its GPU side
        self._default = torch.cuda.default_stream(device_index)
        torch.cuda.set_stream(self._post)
        close_scene = scene.close

        def close():            # torch must not be left on a stream the context is about to destroy
            try:
                torch.cuda.set_stream(self._default)
            except Exception:   # interpreter shutdown: torch may already be gone
                pass
            close_scene()
        scene.close = close

    def set_carry(self, on):
        self.sc.set_carry_accum(on)

    def mix_voices(self, n):
        self.sc.mix_voices_overlapped(n)

    def bus_tensor(self):
        return self._bus

    def collective(self):
        return contextlib.nullcontext()

    def post_process(self, n, run):
        self.sc.post_process_overlapped(n, run)
