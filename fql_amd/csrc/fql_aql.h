// fql_aql.h -- the update program as pre-built AQL packets on the engine's own HSA queues (round 3).
//
// Why: hipGraphLaunch of the three-lane update graph costs the host 260-315 us per update (profiles/r02_graph_knobs.txt: 37 us with
// one queue, so it is the cross-queue bookkeeping, not the packets), as much as the GPU needs for the update itself; the first launch of
// a short window is not hidden at all.  What a launched graph comes to on the device is ~90 64-byte AQL packets on three hardware
// queues.  Here the engine builds those packets ONCE per program - kernel object, grid, LDS size, a kernarg block in device memory - and
// an update is: re-arm a dozen signals, copy the packets into the three rings, ring three doorbells (a few us of host time).
//
//   lanes           one HSA queue per lane; a lane's kernel packets carry the barrier bit (each waits for the lane's previous packet,
//                   what a stream does)
//   cross-lane      the producer's packet carries a completion signal, the consumer lane a barrier-AND packet on it
//   update k -> k+1 lane 0 ends with a barrier-AND on the other lanes' end signals, completion signal DONE[k]; the other lanes start update
//                   k + 1 with a barrier-AND on DONE[k]; lane 0 simply follows it in its queue.  The host waits on DONE.
//   signal sets     8 rotating sets, at most 6 updates in flight (the host blocks on DONE[k - 6] before it re-arms a set)
//   fences          agent scope between packets (what the multi-XCD part needs at a kernel boundary), system scope acquire at the start
//                   and release at the end of an update
//
// The kernels are the ones linked into this library: the gfx950 code object is read back from the library's own .hip_fatbin section and
// loaded through the HSA loader, kernel descriptors are looked up by the names HIP reports for the host stubs (hipKernelNameRefByPtr).
// HIP streams know nothing of these queues: the engine drains them (waits for DONE on the host) before any call that uses HIP on its
// buffers, and drains its HIP stream before it submits here.
#pragma once
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <dlfcn.h>
#include <elf.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <unordered_map>
#include <vector>

#define FQL_AQL_LANES 4
#define FQL_AQL_SETS 8
#define FQL_AQL_INFLIGHT 6

struct AqlError {
    std::string msg;
};
#define AQL_CHECK(expr)                                                                              \
    do {                                                                                             \
        hsa_status_t s_ = (expr);                                                                    \
        if (s_ != HSA_STATUS_SUCCESS) {                                                              \
            const char* m_ = nullptr;                                                                \
            hsa_status_string(s_, &m_);                                                              \
            char buf_[512];                                                                          \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, m_ ? m_ : "?", __FILE__, __LINE__); \
            throw AqlError{buf_};                                                                    \
        }                                                                                            \
    } while (0)

struct AqlKernelInfo {
    uint64_t object = 0;
    uint32_t kernarg_size = 0, group_size = 0, private_size = 0;
};

struct AqlDispatch {   // one recorded launch: what hipLaunchKernelGGL was given
    const void* fn = nullptr;
    unsigned grid[3] = {1, 1, 1}, block[3] = {1, 1, 1};
    uint32_t lds = 0;
    std::vector<uint8_t> args;   // the explicit arguments, laid out as the kernarg segment lays them out
};

// signal references inside a packet template: >= 0 an index into the update's signal set
enum { AQL_SIG_NONE = -1, AQL_SIG_PREV_DONE = -2, AQL_SIG_DONE = -3 };
struct AqlPacket {
    uint32_t w[16] = {};
    int complete = AQL_SIG_NONE;
    int deps[5] = {AQL_SIG_NONE, AQL_SIG_NONE, AQL_SIG_NONE, AQL_SIG_NONE, AQL_SIG_NONE};
    bool barrier = false;   // a barrier-AND packet (deps at words 2 .. 11)
};
struct AqlProgram {
    std::vector<AqlPacket> lane[FQL_AQL_LANES];
    int nsig = 0;             // signals of one set this program uses
    void* kernargs = nullptr; // device memory (hipMalloc), owned by the engine
    bool ok = false;
};

inline uint16_t aql_header(int type, bool barrier, int acquire, int release) {
    return (uint16_t)((type << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                      (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}

struct AqlRuntime {
    bool up = false, failed = false;
    std::string why;                    // why the runtime is not up
    hsa_agent_t agent{};
    hsa_executable_t exe{};
    hsa_code_object_reader_t reader{};
    std::vector<char> image;            // the gfx950 code object (must outlive the reader)
    hsa_queue_t* q[FQL_AQL_LANES] = {};
    std::unordered_map<const void*, AqlKernelInfo> kernels;
    std::vector<hsa_signal_t> sets[FQL_AQL_SETS];
    hsa_signal_t done[FQL_AQL_SETS] = {};
    hsa_signal_t zero{};                // permanently 0: "the update before the first"
    uint64_t submitted = 0, completed = 0;   // updates handed to the queues / known to have finished
    bool hsa_inited = false;

    // ---- bring-up ------------------------------------------------------------------------------------------------------------------
    struct FindCtx { uint32_t domain, bdf; hsa_agent_t out; bool found; };
    static hsa_status_t find_agent(hsa_agent_t a, void* p) {
        FindCtx* c = (FindCtx*)p;
        hsa_device_type_t t;
        if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
        uint32_t bdf = 0, dom = 0;
        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
        hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
        if ((bdf & 0xFFF8u) == (c->bdf & 0xFFF8u) && dom == c->domain) { c->out = a; c->found = true; return HSA_STATUS_INFO_BREAK; }
        return HSA_STATUS_SUCCESS;
    }
    static void queue_error(hsa_status_t st, hsa_queue_t*, void* data) {
        AqlRuntime* r = (AqlRuntime*)data;
        const char* m = nullptr;
        hsa_status_string(st, &m);
        fprintf(stderr, "[fql] AQL queue error: %s\n", m ? m : "?");
        r->failed = true;
    }
    // the gfx950 entry of the clang offload bundle in this library's .hip_fatbin section
    static bool read_code_object(const void* addr_in_lib, const char* isa, std::vector<char>& out, std::string& why) {
        Dl_info di;
        if (!dladdr(addr_in_lib, &di) || !di.dli_fname) { why = "dladdr failed"; return false; }
        std::ifstream f(di.dli_fname, std::ios::binary);
        if (!f) { why = std::string("cannot open ") + di.dli_fname; return false; }
        Elf64_Ehdr eh;
        f.read((char*)&eh, sizeof eh);
        if (!f || std::memcmp(eh.e_ident, ELFMAG, SELFMAG) != 0 || eh.e_shentsize != sizeof(Elf64_Shdr)) { why = "not an ELF64 file"; return false; }
        std::vector<Elf64_Shdr> sh(eh.e_shnum);
        f.seekg((std::streamoff)eh.e_shoff);
        f.read((char*)sh.data(), (std::streamsize)(sh.size() * sizeof(Elf64_Shdr)));
        if (!f || eh.e_shstrndx >= sh.size()) { why = "bad section table"; return false; }
        std::vector<char> names(sh[eh.e_shstrndx].sh_size);
        f.seekg((std::streamoff)sh[eh.e_shstrndx].sh_offset);
        f.read(names.data(), (std::streamsize)names.size());
        for (const Elf64_Shdr& s : sh) {
            if (s.sh_name >= names.size() || std::strcmp(names.data() + s.sh_name, ".hip_fatbin") != 0) continue;
            std::vector<char> fb(s.sh_size);
            f.seekg((std::streamoff)s.sh_offset);
            f.read(fb.data(), (std::streamsize)fb.size());
            if (!f) { why = "short read of .hip_fatbin"; return false; }
            static const char magic[] = "__CLANG_OFFLOAD_BUNDLE__";
            if (fb.size() < 32 || std::memcmp(fb.data(), magic, 24) != 0) { why = ".hip_fatbin is not an uncompressed offload bundle"; return false; }
            uint64_t n;
            std::memcpy(&n, fb.data() + 24, 8);
            size_t o = 32;
            for (uint64_t i = 0; i < n; ++i) {
                if (o + 24 > fb.size()) break;
                uint64_t off, size, tl;
                std::memcpy(&off, fb.data() + o, 8); std::memcpy(&size, fb.data() + o + 8, 8); std::memcpy(&tl, fb.data() + o + 16, 8);
                o += 24;
                if (o + tl > fb.size()) break;
                const std::string triple(fb.data() + o, (size_t)tl);
                o += tl;
                if (triple.find("amdgcn") != std::string::npos && triple.find(isa) != std::string::npos && size > 0 && off + size <= fb.size()) {
                    out.assign(fb.begin() + (std::ptrdiff_t)off, fb.begin() + (std::ptrdiff_t)(off + size));
                    return true;
                }
            }
            why = std::string("no ") + isa + " code object in the bundle";
            return false;
        }
        why = "no .hip_fatbin section";
        return false;
    }

    // hip_device: the device the engine runs on; anchor: any address inside this shared library
    bool init(int hip_device, const void* anchor) {
        try {
            AQL_CHECK(hsa_init());
            hsa_inited = true;
            int bus = 0, dev = 0, dom = 0;
            if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, hip_device) != hipSuccess ||
                hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, hip_device) != hipSuccess ||
                hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, hip_device) != hipSuccess)
                throw AqlError{"PCI address of the HIP device unavailable"};
            FindCtx fc{(uint32_t)dom, (uint32_t)((bus << 8) | (dev << 3)), {}, false};
            hsa_status_t st = hsa_iterate_agents(find_agent, &fc);
            if ((st != HSA_STATUS_SUCCESS && st != HSA_STATUS_INFO_BREAK) || !fc.found) throw AqlError{"no HSA agent at the HIP device's PCI address"};
            agent = fc.out;
            char name[64] = {};
            AQL_CHECK(hsa_agent_get_info(agent, HSA_AGENT_INFO_NAME, name));
            if (!read_code_object(anchor, name, image, why)) throw AqlError{why};
            AQL_CHECK(hsa_code_object_reader_create_from_memory(image.data(), image.size(), &reader));
            AQL_CHECK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
            AQL_CHECK(hsa_executable_load_agent_code_object(exe, agent, reader, nullptr, nullptr));
            AQL_CHECK(hsa_executable_freeze(exe, nullptr));
            AQL_CHECK(hsa_signal_create(0, 0, nullptr, &zero));
            for (int s = 0; s < FQL_AQL_SETS; ++s) AQL_CHECK(hsa_signal_create(0, 0, nullptr, &done[s]));
            up = true;
        } catch (const AqlError& e) {
            why = e.msg;
            up = false;
        }
        return up;
    }
    void ensure_queue(int lane) {
        if (q[lane]) return;
        AQL_CHECK(hsa_queue_create(agent, 4096, HSA_QUEUE_TYPE_SINGLE, queue_error, this, UINT32_MAX, UINT32_MAX, &q[lane]));
    }
    void ensure_signals(int n) {
        for (int s = 0; s < FQL_AQL_SETS; ++s)
            while ((int)sets[s].size() < n) {
                hsa_signal_t sg;
                AQL_CHECK(hsa_amd_signal_create(0, 0, nullptr, HSA_AMD_SIGNAL_AMD_GPU_ONLY, &sg));
                sets[s].push_back(sg);
            }
    }
    const AqlKernelInfo& kernel(const void* host_fn) {
        auto it = kernels.find(host_fn);
        if (it != kernels.end()) return it->second;
        const char* nm = hipKernelNameRefByPtr(host_fn, nullptr);
        if (!nm || !*nm) throw AqlError{"hipKernelNameRefByPtr: no name for a kernel stub"};
        const std::string sym = std::string(nm) + ".kd";
        hsa_executable_symbol_t s;
        hsa_status_t st = hsa_executable_get_symbol_by_name(exe, sym.c_str(), &agent, &s);
        if (st != HSA_STATUS_SUCCESS) throw AqlError{"kernel descriptor not found: " + sym};
        AqlKernelInfo k;
        AQL_CHECK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
        AQL_CHECK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size));
        AQL_CHECK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group_size));
        AQL_CHECK(hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.private_size));
        return kernels.emplace(host_fn, k).first->second;
    }

    // ---- packets -------------------------------------------------------------------------------------------------------------------
    // kernarg block of one dispatch: the explicit arguments, then (code object v5) the hidden block the two kernels that read blockDim /
    // gridDim need: block counts, group sizes, remainders, global offsets, grid dimensions
    static void fill_kernarg(uint8_t* dst, const AqlDispatch& d, const AqlKernelInfo& k) {
        std::memset(dst, 0, k.kernarg_size);
        std::memcpy(dst, d.args.data(), std::min((size_t)k.kernarg_size, d.args.size()));
        const size_t hb = (d.args.size() + 7) & ~(size_t)7;
        if (k.kernarg_size >= hb + 72) {
            uint32_t bc[3] = {d.grid[0], d.grid[1], d.grid[2]};
            uint16_t gs[3] = {(uint16_t)d.block[0], (uint16_t)d.block[1], (uint16_t)d.block[2]};
            std::memcpy(dst + hb, bc, 12);
            std::memcpy(dst + hb + 12, gs, 6);
            const uint16_t dims = d.grid[2] > 1 ? 3 : (d.grid[1] > 1 ? 2 : 1);
            std::memcpy(dst + hb + 64, &dims, 2);
        }
    }
    static AqlPacket kernel_packet(const AqlDispatch& d, const AqlKernelInfo& k, uint64_t kernarg, bool barrier, int acquire, int release) {
        AqlPacket p;
        const uint16_t hdr = aql_header(HSA_PACKET_TYPE_KERNEL_DISPATCH, barrier, acquire, release);
        const uint16_t setup = 3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p.w[0] = (uint32_t)hdr | ((uint32_t)setup << 16);
        p.w[1] = (d.block[0] & 0xFFFFu) | ((d.block[1] & 0xFFFFu) << 16);
        p.w[2] = d.block[2] & 0xFFFFu;
        p.w[3] = d.grid[0] * d.block[0]; p.w[4] = d.grid[1] * d.block[1]; p.w[5] = d.grid[2] * d.block[2];
        p.w[6] = k.private_size;
        p.w[7] = k.group_size + d.lds;
        p.w[8] = (uint32_t)k.object; p.w[9] = (uint32_t)(k.object >> 32);
        p.w[10] = (uint32_t)kernarg; p.w[11] = (uint32_t)(kernarg >> 32);
        return p;
    }
    static AqlPacket barrier_packet(bool barrier, int acquire, int release) {
        AqlPacket p;
        p.barrier = true;
        p.w[0] = aql_header(HSA_PACKET_TYPE_BARRIER_AND, barrier, acquire, release);
        return p;
    }

    // ---- submit / wait -------------------------------------------------------------------------------------------------------------
    void wait_update(uint64_t k) {   // update number k (0-based) has finished on the device
        if (k < completed) return;
        if (k >= submitted) throw AqlError{"wait for an update that was never submitted"};
        if (submitted - k > FQL_AQL_SETS) throw AqlError{"wait for an update whose signal set was re-armed"};
        hsa_signal_t sg = done[k % FQL_AQL_SETS];
        for (int tries = 0; tries < 30; ++tries) {   // 30 s in all: a stuck queue is reported, not waited for
            if (hsa_signal_wait_scacquire(sg, HSA_SIGNAL_CONDITION_EQ, 0, 1000000000ull, HSA_WAIT_STATE_ACTIVE) == 0) { completed = k + 1; return; }
            if (failed) break;
        }
        failed = true;
        throw AqlError{"AQL queue error (see stderr), or an update did not finish within 30 s"};
    }
    void drain() {
        if (submitted > completed) wait_update(submitted - 1);
    }
    void submit(const AqlProgram& P) {
        if (failed) throw AqlError{"AQL runtime is in a failed state"};
        const uint64_t k = submitted;
        const int s = (int)(k % FQL_AQL_SETS);
        if (k >= FQL_AQL_INFLIGHT) wait_update(k - FQL_AQL_INFLIGHT);
        for (int i = 0; i < P.nsig; ++i) hsa_signal_store_relaxed(sets[s][i], 1);
        hsa_signal_store_relaxed(done[s], 1);
        const hsa_signal_t prev = k == 0 ? zero : done[(k - 1) % FQL_AQL_SETS];
        auto resolve = [&](int ref) -> uint64_t {
            if (ref == AQL_SIG_NONE) return 0;
            if (ref == AQL_SIG_PREV_DONE) return prev.handle;
            if (ref == AQL_SIG_DONE) return done[s].handle;
            return sets[s][ref].handle;
        };
        for (int l = FQL_AQL_LANES - 1; l >= 0; --l) {
            const std::vector<AqlPacket>& pk = P.lane[l];
            if (pk.empty()) continue;
            hsa_queue_t* Q = q[l];
            const uint64_t n = pk.size();
            const uint64_t idx = hsa_queue_load_write_index_relaxed(Q);
            for (int spins = 0; idx + n - hsa_queue_load_read_index_scacquire(Q) > Q->size; ++spins) {
                if (failed || spins > (1 << 28)) { failed = true; throw AqlError{"AQL ring stayed full"}; }
                __builtin_ia32_pause();
            }
            const uint64_t mask = Q->size - 1;
            for (uint64_t i = 0; i < n; ++i) {
                const AqlPacket& t = pk[i];
                uint32_t* slot = (uint32_t*)Q->base_address + ((idx + i) & mask) * 16;
                uint32_t w[16];
                std::memcpy(w, t.w, sizeof w);
                if (t.barrier)
                    for (int j = 0; j < 5; ++j) { const uint64_t h = resolve(t.deps[j]); w[2 + 2 * j] = (uint32_t)h; w[3 + 2 * j] = (uint32_t)(h >> 32); }
                const uint64_t c = resolve(t.complete);
                w[14] = (uint32_t)c; w[15] = (uint32_t)(c >> 32);
                for (int j = 1; j < 16; ++j) slot[j] = w[j];
                __atomic_store_n(slot, w[0], __ATOMIC_RELEASE);   // the header last: the packet processor may take the packet from here on
            }
            // one doorbell per contiguous run: a batch that wraps the ring is rung in two parts (a queue intercepted by a tool - rocprofv3's kernel
            // trace - was handed the whole batch as one array and read past the end of the ring: SIGSEGV on the first wrap, 96 updates in)
            const uint64_t to_end = Q->size - (idx & mask);
            if (n > to_end) {
                hsa_queue_store_write_index_relaxed(Q, idx + to_end);
                hsa_signal_store_screlease(Q->doorbell_signal, (hsa_signal_value_t)(idx + to_end - 1));
            }
            hsa_queue_store_write_index_relaxed(Q, idx + n);
            hsa_signal_store_screlease(Q->doorbell_signal, (hsa_signal_value_t)(idx + n - 1));
        }
        ++submitted;
    }

    void shutdown() {
        if (up) { try { drain(); } catch (...) {} }
        for (auto& Q : q) if (Q) { hsa_queue_destroy(Q); Q = nullptr; }
        for (auto& v : sets) { for (auto sg : v) hsa_signal_destroy(sg); v.clear(); }
        if (up) {
            for (auto sg : done) hsa_signal_destroy(sg);
            hsa_signal_destroy(zero);
            hsa_executable_destroy(exe);
            hsa_code_object_reader_destroy(reader);
        }
        up = false;
        if (hsa_inited) { hsa_shut_down(); hsa_inited = false; }
    }
};
