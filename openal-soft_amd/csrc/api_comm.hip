// The sharded update's one exchange: RCCL and host-staged transports, oalgpu_comm_*.
#include "api_context.hpp"

namespace {
// ---- RCCL, resolved at run time: a single-GPU host never needs the library, and a process that already
// carries an RCCL (torch's) must use THAT instance rather than a second copy
struct RcclApi {
    ncclResult_t (*getUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*commInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*getErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*commCount)(const ncclComm_t, int*) = nullptr;
    bool ok = false;
    std::string why;
};

RcclApi &Rccl()
{
    static RcclApi api = []
    {
        RcclApi a;
        void *h = nullptr;
        if(dlsym(RTLD_DEFAULT, "ncclCommInitRank")) h = RTLD_DEFAULT;       // already in the process
        else
        {
            for(const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        }
        if(!h)
        {   // dlerror() clears the pending message: read it once
            const char *e = dlerror();
            a.why = std::string("librccl.so not found: ") + (e ? e : "");
            return a;
        }
        a.getUniqueId = reinterpret_cast<decltype(a.getUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        a.commInitRank = reinterpret_cast<decltype(a.commInitRank)>(dlsym(h, "ncclCommInitRank"));
        a.commDestroy = reinterpret_cast<decltype(a.commDestroy)>(dlsym(h, "ncclCommDestroy"));
        a.reduce = reinterpret_cast<decltype(a.reduce)>(dlsym(h, "ncclReduce"));
        a.getErrorString = reinterpret_cast<decltype(a.getErrorString)>(dlsym(h, "ncclGetErrorString"));
        a.commCount = reinterpret_cast<decltype(a.commCount)>(dlsym(h, "ncclCommCount"));
        a.ok = a.getUniqueId && a.commInitRank && a.commDestroy && a.reduce;
        if(!a.ok) a.why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclReduce";
        return a;
    }();
    return api;
}

int FailRccl(const char *what, ncclResult_t r)
{
    const RcclApi &a = Rccl();
    return Fail(OALGPU_ERR_HIP, std::string(what) + ": " + (a.getErrorString ? a.getErrorString(r) : "RCCL error"));
}

} // namespace

namespace {

struct RcclTransport final : BusTransport {
    ncclComm_t comm{nullptr};
    ~RcclTransport() override { if(comm) (void)Rccl().commDestroy(comm); }
    int reduceToRoot(oalgpu_context *c, hipStream_t s) override
    {
        const ncclResult_t r = Rccl().reduce(c->L.bus, c->L.bus, BusFloats(c->L), ncclFloat32, ncclSum, 0, comm, s);
        if(r != ncclSuccess) return FailRccl("ncclReduce", r);
        return OALGPU_OK;
    }
    int ranks() const override
    {
        int n = 0;
        return (Rccl().commCount && Rccl().commCount(comm, &n) == ncclSuccess) ? n : -1;
    }
    const char *kind() const override { return "rccl"; }
};

__global__ void AddBusKernel(float *__restrict__ bus, const float *__restrict__ add, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) bus[i] = bus[i] + add[i];
}

// Shared-memory ring: kSlots updates deep, so that ranks may run that far ahead of rank 0 (the pipelined update
// never synchronises with the host).  produced[r] = updates of rank r whose block is in the ring; consumed =
// updates rank 0 has summed.  Host functions in stream order (hipLaunchHostFunc) move the data; they only touch
// host memory.
struct HostTransport final : BusTransport {
    static constexpr uint32_t kSlots = 4, kMaxWorld = 16;
    struct Header {
        std::atomic<uint32_t> magic;
        uint32_t world, floats;
        std::atomic<uint64_t> produced[kMaxWorld];
        std::atomic<uint64_t> consumed;
        std::atomic<uint32_t> failed;
        std::atomic<uint64_t> hello[kMaxWorld], ack[kMaxWorld];     // the attach handshake (oalgpu_comm_init_host)
    };
    std::string name;
    int fd{-1}, rank{0}, world{1};
    size_t bytes{0}, floats{0};
    Header *hdr{nullptr};
    float *ring{nullptr};                          // [rank][slot][floats]
    float *pinned[kSlots]{};                       // this rank's staging: D2H target (rank > 0), H2D source (rank 0)
    DevBuf<float> devSum;                          // rank 0: the other ranks' sum on the device
    // Host functions of one stream run in stream order, so each side counts the updates it has EXECUTED itself: a
    // sequence number handed over through a reusable host slot would be overwritten by a host that is kSlots or
    // more updates ahead of its stream (nothing throttles it: oalgpu_mix_update never synchronises)
    uint64_t executed{0};
    uint64_t submitted{0};                         // updates enqueued by the host (selects the staging slot)

    float *slot(int r, uint64_t q) const { return ring + (size_t(r) * kSlots + size_t(q % kSlots)) * floats; }
    int ranks() const override { return hdr ? int(hdr->world) : world; }
    const char *kind() const override { return "host"; }

    static bool WaitFor(const std::function<bool()> &ok, std::atomic<uint32_t> &failed)
    {
        const auto t0 = std::chrono::steady_clock::now();
        for(uint32_t spins = 0; !ok(); ++spins)
        {
            if(failed.load(std::memory_order_relaxed)) return false;
            if(spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(20));
            if((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
            { failed.store(1u); return false; }
        }
        return true;
    }
    static void Produce(void *p)
    {   // rank > 0: this update's block (already in pinned memory) into the ring
        HostTransport *t = static_cast<HostTransport*>(p);
        const uint64_t q = t->executed++;
        if(!WaitFor([&] { return q < t->hdr->consumed.load(std::memory_order_acquire) + kSlots; }, t->hdr->failed)) return;
        std::memcpy(t->slot(t->rank, q), t->pinned[q % kSlots], t->floats * sizeof(float));
        t->hdr->produced[t->rank].store(q + 1, std::memory_order_release);
    }
    static void Gather(void *p)
    {   // rank 0: the other ranks' blocks of this update, summed in rank order
        HostTransport *t = static_cast<HostTransport*>(p);
        const uint64_t q = t->executed++;
        float *dst = t->pinned[q % kSlots];
        for(int r = 1; r < t->world; ++r)
        {
            if(!WaitFor([&] { return t->hdr->produced[r].load(std::memory_order_acquire) > q; }, t->hdr->failed))
            { std::memset(dst, 0, t->floats * sizeof(float)); return; }
            const float *src = t->slot(r, q);
            if(r == 1) std::memcpy(dst, src, t->floats * sizeof(float));
            else for(size_t i = 0; i < t->floats; ++i) dst[i] += src[i];
        }
        t->hdr->consumed.store(q + 1, std::memory_order_release);
    }
    int reduceToRoot(oalgpu_context *c, hipStream_t s) override
    {
        if(hdr->failed.load()) return Fail(OALGPU_ERR_HIP, "host transport: a rank timed out waiting for its peers");
        // (the pinned slot of update q is next written by the copy of update q + kSlots, which the stream runs behind
        // Produce / the H2D copy of update q: stream order alone keeps the staging slots apart)
        const uint64_t q = submitted++;
        if(rank != 0)
        {
            HIP_TRY(hipMemcpyAsync(pinned[q % kSlots], c->L.bus, bytes, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipLaunchHostFunc(s, Produce, this));
            return OALGPU_OK;
        }
        if(world == 1) return OALGPU_OK;
        HIP_TRY(hipLaunchHostFunc(s, Gather, this));
        HIP_TRY(hipMemcpyAsync(devSum.p, pinned[q % kSlots], bytes, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(AddBusKernel, dim3(uint32_t((floats + 255) / 256)), dim3(256), 0, s, c->L.bus, devSum.p, uint32_t(floats));
        HIP_TRY(hipGetLastError());
        return OALGPU_OK;
    }
    ~HostTransport() override
    {
        for(float *p : pinned) if(p) (void)hipHostFree(p);
        if(hdr) munmap(hdr, sizeof(Header) + size_t(world) * kSlots * bytes);
        if(fd >= 0) close(fd);
        if(rank == 0 && !name.empty()) shm_unlink(name.c_str());
    }
};

} // namespace

int CommReduceBus(oalgpu_context *c, hipStream_t s)
{
    if(!c->comm) return OALGPU_OK;
    return c->comm->reduceToRoot(c, s);
}

extern "C" {

/* ---- multi-GPU: voices shard over the GPUs of a node, one context per GPU and process ----------------
 * Rank 0 calls oalgpu_comm_unique_id and hands the 128 bytes to the other ranks by whatever means the
 * host has (a file, MPI, torch.distributed); every rank then calls oalgpu_comm_init on its context.
 * From then on oalgpu_mix_update / oalgpu_mix_voices sum-reduce the bus block to rank 0 (ncclReduce over
 * xGMI, issued by the library on the stream that produced the buses -- the context's post stream in the
 * pipelined path, so it runs beside the next update's voice kernel), and only rank 0 -- the one rank
 * that carries the HRTF accumulator tail -- runs the effect slots and the post-process. */
int oalgpu_comm_unique_id(void *out, size_t size)
{
    if(!out || size < sizeof(ncclUniqueId)) return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_unique_id: 128 bytes needed");
    RcclApi &a = Rccl();
    if(!a.ok) return Fail(OALGPU_ERR_NO_DEVICE, a.why);
    ncclUniqueId id;
    const ncclResult_t r = a.getUniqueId(&id);
    if(r != ncclSuccess) return FailRccl("ncclGetUniqueId", r);
    std::memcpy(out, &id, sizeof(id));
    return OALGPU_OK;
}

int oalgpu_comm_init(oalgpu_context *c, const void *unique_id, size_t size, int rank, int world)
{
    if(!c || !unique_id || size < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init: bad arguments");
    if(c->comm) return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init: the context already has a communicator");
    if(!c->cbVoices.empty()) return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init: not on a context with callback sources");
    RcclApi &a = Rccl();
    if(!a.ok) return Fail(OALGPU_ERR_NO_DEVICE, a.why);
    if(int rc = UseCtx(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    auto t = std::make_unique<RcclTransport>();
    const ncclResult_t r = a.commInitRank(&t->comm, world, id, rank);
    if(r != ncclSuccess) return FailRccl("ncclCommInitRank", r);
    c->comm = t.release(); c->commRank = rank; c->commWorld = world;
    c->carryAccum = rank == 0;          // exactly one rank continues the carried HRTF accumulator
    return OALGPU_OK;
}

/* The same sharded update over the host-staged transport: `name` = a POSIX shared-memory object name ("/..."),
 * the same on every rank; rank 0 creates it, the others attach (they wait for it to appear).  For ranks that RCCL
 * cannot connect -- several processes on one GPU. */
int oalgpu_comm_init_host(oalgpu_context *c, const char *name, int rank, int world)
{
    if(!c || !name || name[0] != '/' || world < 1 || world > int(HostTransport::kMaxWorld) || rank < 0 || rank >= world)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init_host: bad arguments");
    if(c->comm) return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init_host: the context already has a communicator");
    if(!c->cbVoices.empty()) return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init_host: not on a context with callback sources");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    auto t = std::make_unique<HostTransport>();
    t->name = name; t->rank = rank; t->world = world;
    t->floats = BusFloats(c->L); t->bytes = t->floats * sizeof(float);
    const size_t total = sizeof(HostTransport::Header) + size_t(world) * HostTransport::kSlots * t->bytes;
    using clk = std::chrono::steady_clock;
    const auto deadline = clk::now() + std::chrono::seconds(60);
    // everything that can fail on this side comes first: a rank never announces itself and then falls over an allocation
    for(float *&p : t->pinned) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p), t->bytes, hipHostMallocDefault));
    if(rank == 0)
    {
        HIP_TRY(t->devSum.alloc(t->floats));
        shm_unlink(name);
        t->fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if(t->fd < 0 || ftruncate(t->fd, off_t(total)) != 0) return Fail(OALGPU_ERR_HIP, std::string("shm_open/ftruncate ") + name + " failed");
        void *m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, t->fd, 0);
        if(m == MAP_FAILED) return Fail(OALGPU_ERR_HIP, "mmap of the shared segment failed");
        t->hdr = static_cast<HostTransport::Header*>(m);
        t->ring = reinterpret_cast<float*>(static_cast<char*>(m) + sizeof(HostTransport::Header));
        // (a fresh segment is zero-filled: produced, consumed, failed, hello, ack start at 0)
        t->hdr->world = uint32_t(world); t->hdr->floats = uint32_t(t->floats);
        t->hdr->magic.store(0x0a16b05u, std::memory_order_release);
        // The attach is a handshake, so that no rank can sit on a segment a crashed earlier run left under the same
        // name (rank 0 only unlinks in its destructor): every other rank writes a token of its own into hello[r] and
        // trusts the segment only once THIS rank 0 has echoed it into ack[r]; a stale segment never answers.
        for(int r = 1; r < world; ++r)
        {
            uint64_t tok = 0;
            while((tok = t->hdr->hello[r].load(std::memory_order_acquire)) == 0)
            {
                if(clk::now() > deadline) return Fail(OALGPU_ERR_HIP, "oalgpu_comm_init_host: a rank did not attach within 60 s");
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            t->hdr->ack[r].store(tok, std::memory_order_release);
        }
    }
    else
    {
        const uint64_t token = ((uint64_t(getpid()) << 32) ^ uint64_t(clk::now().time_since_epoch().count()) ^ (uint64_t(rank) << 56)) | 1ull;
        bool attached = false;
        std::string why = "rank 0's segment did not appear";
        while(!attached && clk::now() < deadline)
        {
            t->fd = shm_open(name, O_RDWR, 0600);
            struct stat st{};
            if(t->fd >= 0 && (fstat(t->fd, &st) != 0 || size_t(st.st_size) < total)) { close(t->fd); t->fd = -1; }
            if(t->fd < 0) { std::this_thread::sleep_for(std::chrono::milliseconds(10)); continue; }
            void *m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, t->fd, 0);
            if(m == MAP_FAILED) return Fail(OALGPU_ERR_HIP, "mmap of the shared segment failed");
            auto *hdr = static_cast<HostTransport::Header*>(m);
            const auto patience = clk::now() + std::chrono::seconds(3);     // a live rank 0 answers within milliseconds
            bool said = false;
            while(clk::now() < patience && clk::now() < deadline)
            {
                if(hdr->magic.load(std::memory_order_acquire) == 0x0a16b05u)
                {
                    if(hdr->world != uint32_t(world) || hdr->floats != uint32_t(t->floats)) { why = "the ranks' contexts differ (world size or bus block)"; break; }
                    if(!said) { hdr->hello[rank].store(token, std::memory_order_release); said = true; }
                    if(hdr->ack[rank].load(std::memory_order_acquire) == token) { attached = true; break; }
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if(attached)
            {
                t->hdr = hdr;
                t->ring = reinterpret_cast<float*>(static_cast<char*>(m) + sizeof(HostTransport::Header));
            }
            else
            {   // nobody answered: a segment left behind by an earlier run (rank 0 replaces it), or a mismatch
                munmap(m, total); close(t->fd); t->fd = -1;
                if(why.find("differ") != std::string::npos) return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init_host: " + why);
                why = "no live rank 0 answered on the segment";
            }
        }
        if(!attached) return Fail(OALGPU_ERR_HIP, std::string("shm_open ") + name + ": " + why);
    }
    c->comm = t.release(); c->commRank = rank; c->commWorld = world;
    c->carryAccum = rank == 0;
    return OALGPU_OK;
}

/* what the context's exchange looks like from the inside: this rank, the world it was given, and the ranks the transport itself
 * counts (RCCL: ncclCommCount of the communicator the library created; -1: the library's RCCL has no such call) */
int oalgpu_comm_info(oalgpu_context *c, int *rank, int *world, int *transport_ranks, char *kind, size_t kind_size)
{
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(rank) *rank = c->commRank;
    if(world) *world = c->commWorld;
    if(transport_ranks) *transport_ranks = c->comm ? c->comm->ranks() : 1;
    if(kind && kind_size) { std::snprintf(kind, kind_size, "%s", c->comm ? c->comm->kind() : "none"); }
    return OALGPU_OK;
}

int oalgpu_comm_destroy(oalgpu_context *c)
{
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(!c->comm) return OALGPU_OK;
    if(int rc = oalgpu_sync(c)) return rc;
    delete c->comm;
    c->comm = nullptr; c->commRank = 0; c->commWorld = 1; c->carryAccum = true;
    return OALGPU_OK;
}

} // extern "C"
