"""Voice sharding across the GPUs of one node: voices are independent given
their parameters, so every rank owns a shard of them and mixes it into PARTIAL buses; the only
exchange is one sum-reduce of the bus block [dry+real lines | wet buses | HrtfAccumData] to rank 0
per update (~41 KB over xGMI), after which rank 0 -- the only rank that carries the HRTF accumulator
tail between updates -- runs the effects and the HRTF post-process (alc/alu.cpp:2209-2257, :289-298)
on the summed buses.

On GPUs the exchange is issued by the LIBRARY (oalgpu_comm_init + oalgpu_mix_update: ncclReduce on the
context's post stream; nothing of it lives here or in torch).  This module holds what is left for a
host to decide -- which voices go to which rank -- and ``ShardedMixer``, the same scheme spelled out
over a generic ``engine`` + torch.distributed, which tests/test_multi_rank.py runs on the CPU oracle
over gloo (world_size 2) to check the scheme itself against an unsharded scene:
    engine.set_carry(bool) / mix_voices(n) / bus_tensor() -> torch tensor aliasing the bus block /
    post_process(n, run)"""


def shard_range(total_voices, rank, world):
    """Contiguous, near-equal shards of a scene whose voices all cost the same: rank r owns [lo, hi)."""
    base, extra = divmod(total_voices, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def voice_cost(hrtf, resampler_taps=24, active_sends=0, filtered=False):
    """Relative cost of one voice-update (SURVEY.md 8d flops: resample taps*4, HRTF 64*2*2+4 per sample,
    a dry-line mix ~2 per line, +1 class per active send, a biquad pair 18): the weights of
    ``weighted_shards``.  Only ratios matter."""
    cost = resampler_taps * 4.0 + (260.0 if hrtf else 10.0)
    cost += active_sends * (8.0 + 18.0 / 3.0)
    if filtered:
        cost += 18.0
    return cost


def weighted_shards(costs, world, rank0_extra=0.0):
    """Static assignment of voices to ranks by cost: longest-processing-time first -- the voices
    in order of falling cost, each onto the rank that carries the least so far (ties: the lowest rank).  Loads end
    within one voice's cost of each other; since voices of one class cost the same, every rank also ends up with
    close to the same number of each class, but that is a consequence, not a guarantee.  ``rank0_extra`` is work
    only rank 0 has (effect slots, post-process), in the units of ``costs``: rank 0 starts that far ahead.
    Returns one sorted voice-index list per rank."""
    order = sorted(range(len(costs)), key=lambda v: (-costs[v], v))
    load = [float(rank0_extra)] + [0.0] * (world - 1)
    shards = [[] for _ in range(world)]
    for v in order:                       # longest-processing-time first onto the least loaded rank
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(v)
        load[r] += costs[v]
    return [sorted(s) for s in shards]


class ShardedMixer:
    def __init__(self, engine, dist=None, rank=0, world=1):
        self.engine, self.dist, self.rank, self.world = engine, dist, rank, world
        engine.set_carry(rank == 0)

    def update(self, samples_to_do):
        e = self.engine
        e.mix_voices(samples_to_do)
        if self.dist is not None and (self.world > 1 or getattr(e, "always_reduce", False)):
            self.dist.reduce(e.bus_tensor(), dst=0, op=self.dist.ReduceOp.SUM)
        e.post_process(samples_to_do, self.rank == 0)
