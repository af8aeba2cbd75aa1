// jit.hip -- see jit.hpp.
#include "jit.hpp"

#include <dlfcn.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace tinympc_amd {
namespace {

#include "_gen/kernel_src.inc"      // static const char kAdmmKernelSrc[] = R"(...admm_kernel.hip.h...)";

// the handful of hipRTC entry points, resolved at first use
struct Rtc {
    typedef void* Program;
    int (*create)(Program*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    int (*add_name)(Program, const char*) = nullptr;
    int (*compile)(Program, int, const char* const*) = nullptr;
    int (*log_size)(Program, size_t*) = nullptr;
    int (*log)(Program, char*) = nullptr;
    int (*lowered)(Program, const char*, const char**) = nullptr;
    int (*code_size)(Program, size_t*) = nullptr;
    int (*code)(Program, char*) = nullptr;
    int (*destroy)(Program*) = nullptr;
    int (*version)(int*, int*) = nullptr;
    bool ok = false;
};

Rtc& rtc() {
    static Rtc r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    void* h = nullptr;
    for (const char* n : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"})
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return r;
    auto sym = [&](const char* n) { return dlsym(h, n); };
    r.create = reinterpret_cast<decltype(r.create)>(sym("hiprtcCreateProgram"));
    r.add_name = reinterpret_cast<decltype(r.add_name)>(sym("hiprtcAddNameExpression"));
    r.compile = reinterpret_cast<decltype(r.compile)>(sym("hiprtcCompileProgram"));
    r.log_size = reinterpret_cast<decltype(r.log_size)>(sym("hiprtcGetProgramLogSize"));
    r.log = reinterpret_cast<decltype(r.log)>(sym("hiprtcGetProgramLog"));
    r.lowered = reinterpret_cast<decltype(r.lowered)>(sym("hiprtcGetLoweredName"));
    r.code_size = reinterpret_cast<decltype(r.code_size)>(sym("hiprtcGetCodeSize"));
    r.code = reinterpret_cast<decltype(r.code)>(sym("hiprtcGetCode"));
    r.destroy = reinterpret_cast<decltype(r.destroy)>(sym("hiprtcDestroyProgram"));
    r.version = reinterpret_cast<decltype(r.version)>(sym("hiprtcVersion"));
    r.ok = r.create && r.add_name && r.compile && r.log_size && r.log && r.lowered && r.code_size && r.code && r.destroy;
    return r;
}

// ---- compiled code objects: per process by instantiation name, optionally on disk ---------------------------------
struct Code { std::vector<char> blob; std::string lowered, err; bool from_disk = false; };
struct Entry { hipFunction_t fn = nullptr; std::string err; };
std::map<std::string, Code> g_code;                     // by instantiation name
std::map<std::string, Entry> g_cache;                   // by instantiation name @ device (a module belongs to one device)
std::mutex g_mu;

uint64_t fnv(uint64_t h, const void* p, size_t n) {
    const unsigned char* c = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    return h;
}
const char* const kOpts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
const char kMagic[8] = {'T', 'M', 'P', 'C', 'J', 'I', 'T', '1'};

// TINYMPC_AMD_JIT_CACHE=<directory>: file of one instantiation = hash of (both kernel headers, name, options, hipRTC version)
std::string disk_path(const std::string& name) {
    const char* dir = getenv("TINYMPC_AMD_JIT_CACHE");
    if (!dir || !*dir) return "";
    uint64_t h = 1469598103934665603ull;
    h = fnv(h, kAdmmKernelSrc, sizeof(kAdmmKernelSrc));
    h = fnv(h, kTileKernelSrc, sizeof(kTileKernelSrc));
    h = fnv(h, name.data(), name.size());
    for (const char* o : kOpts) h = fnv(h, o, strlen(o));
    int ver[2] = {0, 0};
    if (rtc().version) rtc().version(&ver[0], &ver[1]);
    h = fnv(h, ver, sizeof(ver));
    char file[40];
    snprintf(file, sizeof(file), "/tinympc_amd_%016llx.co", (unsigned long long)h);
    return std::string(dir) + file;
}

// The PREBUILT store (round 6): code objects compiled at BUILD time by the build machine's toolchain (tinympc_amd.build() ->
// tiny_jit_prebuild for every name of csrc/jit_prebuilt.txt), in <directory of the library>/jit_prebuilt/ (TINYMPC_AMD_JIT_PREBUILT=<dir>
// overrides, "0" switches it off).  Keyed by the kernel headers, the name and the options -- NOT by the hipRTC version of the process:
// a process that has loaded another ROCm (PyTorch's wheel brings its own libhiprtc / libamd_comgr, which serve every later dlopen by
// soname) then runs the build's code instead of what that older compiler makes of the same source (measured: the per-instance-data form
// of (20,8,10), 5.7 ms compiled by ROCm 7.2, 8.1 ms by the 7.0 compiler inside torch 2.10 -- profiles/r06_jit_compiler_probe.md), and
// the first launch of a listed form costs a file read instead of seconds of compilation.
std::string prebuilt_dir() {
    if (const char* e = getenv("TINYMPC_AMD_JIT_PREBUILT")) return strcmp(e, "0") ? std::string(e) : std::string();
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void*>(&prebuilt_dir), &info) || !info.dli_fname) return "";
    std::string path(info.dli_fname);
    const size_t cut = path.find_last_of('/');
    return (cut == std::string::npos ? std::string(".") : path.substr(0, cut)) + "/jit_prebuilt";
}
std::string prebuilt_path(const std::string& dir, const std::string& name) {
    if (dir.empty()) return "";
    uint64_t h = 1469598103934665603ull;
    h = fnv(h, kAdmmKernelSrc, sizeof(kAdmmKernelSrc));
    h = fnv(h, kTileKernelSrc, sizeof(kTileKernelSrc));
    h = fnv(h, name.data(), name.size());
    for (const char* o : kOpts) h = fnv(h, o, strlen(o));
    char file[48];
    snprintf(file, sizeof(file), "/tinympc_amd_pre_%016llx.co", (unsigned long long)h);
    return dir + file;
}

// file = magic | u32 len(lowered) | u64 len(code) | lowered | code | u64 fnv(code); anything unexpected = not cached
bool disk_load(const std::string& path, Code* c) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[8];
    uint32_t nl = 0;
    uint64_t nc = 0, sum = 0;
    bool ok = fread(magic, 1, 8, f) == 8 && !memcmp(magic, kMagic, 8) && fread(&nl, 4, 1, f) == 1 && fread(&nc, 8, 1, f) == 1 &&
              nl > 0 && nl < 4096 && nc > 0 && nc < (64u << 20);
    if (ok) {
        c->lowered.resize(nl);
        c->blob.resize(nc);
        ok = fread(&c->lowered[0], 1, nl, f) == nl && fread(c->blob.data(), 1, nc, f) == nc && fread(&sum, 8, 1, f) == 1 &&
             sum == fnv(1469598103934665603ull, c->blob.data(), nc);
    }
    fclose(f);
    if (!ok) { c->lowered.clear(); c->blob.clear(); }
    return ok;
}
void disk_store(const std::string& path, const Code& c) {                // best effort; rename makes it atomic for concurrent ranks
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return;
    const uint32_t nl = (uint32_t)c.lowered.size();
    const uint64_t nc = c.blob.size(), sum = fnv(1469598103934665603ull, c.blob.data(), c.blob.size());
    const bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(&nl, 4, 1, f) == 1 && fwrite(&nc, 8, 1, f) == 1 && fwrite(c.lowered.data(), 1, nl, f) == nl &&
                    fwrite(c.blob.data(), 1, nc, f) == nc && fwrite(&sum, 8, 1, f) == 1;
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
}

Code compile(const std::string& name_s, const bool tile) {
    Code c;
    const char* name = name_s.c_str();
    const std::string path = disk_path(name_s);
    if (!path.empty() && disk_load(path, &c)) { c.from_disk = true; return c; }
    if (!getenv("TINYMPC_AMD_JIT_DEFINES")) {         // (experiment builds define macros the prebuilt objects were not made with)
        const std::string pre = prebuilt_path(prebuilt_dir(), name_s);
        if (!pre.empty() && disk_load(pre, &c)) { c.from_disk = true; return c; }
    }
    Rtc& R = rtc();
    if (!R.ok) { c.err = "libhiprtc is not available"; return c; }
    // hipRTC brings its own runtime header: the two system includes of the kernel header are dropped
    std::string hdr(kAdmmKernelSrc);
    for (const char* inc : {"#include <hip/hip_runtime.h>", "#include <stdint.h>"}) {
        const size_t p = hdr.find(inc);
        if (p != std::string::npos) hdr.replace(p, std::string(inc).size(), "");
    }
    std::string src = tile ? "#include \"tile_kernel.hip.h\"\n" : "#include \"admm_kernel.hip.h\"\n";
    {   // the fused sweep steps of admm_kernel.hip.h are spelled out for ONE (nx, nu) pair per compilation: this one's (the tile
        // kernel uses them too when an instance is one row wide, W = 1: nx + nu <= 16)
        int fnx = 0, fnu = 0;
        const size_t lt = name_s.find('<');
        if (lt != std::string::npos && sscanf(name_s.c_str() + lt + 1, "%d , %d", &fnx, &fnu) == 2 && fnx > 0 && fnu > 0 && fnx + fnu <= 16)
            src = "#define TINYMPC_FUSED_NX " + std::to_string(fnx) + "\n#define TINYMPC_FUSED_NU " + std::to_string(fnu) + "\n" + src;
    }
    // experiments: TINYMPC_AMD_JIT_DEFINES="NAME=value;NAME2=value2" puts those macros in front of the kernel source (not part of
    // the disk cache's key: do not combine the two)
    if (const char* defs = getenv("TINYMPC_AMD_JIT_DEFINES")) {
        std::string d(defs), pre;
        size_t a = 0;
        while (a < d.size()) {
            size_t e = d.find(';', a);
            if (e == std::string::npos) e = d.size();
            std::string one = d.substr(a, e - a);
            const size_t eq = one.find('=');
            if (!one.empty()) pre += "#define " + (eq == std::string::npos ? one : one.substr(0, eq) + " " + one.substr(eq + 1)) + "\n";
            a = e + 1;
        }
        src = pre + src;
    }
    const char* hn[] = {"admm_kernel.hip.h", "tile_kernel.hip.h"};
    const char* hs[] = {hdr.c_str(), kTileKernelSrc};
    Rtc::Program prog = nullptr;
    if (R.create(&prog, src.c_str(), "tinympc_amd_jit.hip", 2, hs, hn) != 0) { c.err = "hiprtcCreateProgram failed"; return c; }
    R.add_name(prog, name);
    const int rc = R.compile(prog, 3, kOpts);
    if (rc != 0) {
        size_t n = 0;
        R.log_size(prog, &n);
        std::string log(n, '\0');
        if (n) R.log(prog, &log[0]);
        c.err = "hipRTC compilation of " + std::string(name) + " failed: " + log.substr(0, 400);
        R.destroy(&prog);
        return c;
    }
    const char* low = nullptr;
    size_t cs = 0;
    if (R.lowered(prog, name, &low) != 0 || R.code_size(prog, &cs) != 0 || cs == 0) { c.err = "hipRTC produced no code"; R.destroy(&prog); return c; }
    c.blob.resize(cs);
    R.code(prog, c.blob.data());
    c.lowered = low;
    R.destroy(&prog);
    if (!path.empty()) disk_store(path, c);
    return c;
}

const Code& code_for(const std::string& name, const bool tile) {          // g_mu held
    auto it = g_code.find(name);
    if (it == g_code.end()) it = g_code.emplace(name, compile(name, tile)).first;
    return it->second;
}

Entry build(const std::string& name, const bool tile) {
    Entry e;
    const Code& c = code_for(name, tile);
    if (c.blob.empty()) { e.err = c.err; return e; }
    hipModule_t mod = nullptr;
    if (hipModuleLoadData(&mod, c.blob.data()) != hipSuccess) { (void)hipGetLastError(); e.err = "hipModuleLoadData failed"; return e; }
    if (hipModuleGetFunction(&e.fn, mod, c.lowered.c_str()) != hipSuccess) { (void)hipGetLastError(); e.fn = nullptr; e.err = "kernel symbol not found in the compiled module"; }
    return e;                                            // the module stays loaded for the life of the process
}

}  // namespace

static hipFunction_t get(const std::string& name, const bool tile, std::string* err) {
    std::lock_guard<std::mutex> lk(g_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const std::string key = name + "@" + std::to_string(dev);
    auto it = g_cache.find(key);
    if (it == g_cache.end()) it = g_cache.emplace(key, build(name, tile)).first;
    if (!it->second.fn && err) *err = it->second.err;
    return it->second.fn;
}

long jit_compile_only(const char* instantiation, int* from_disk, std::string* err) {
    const std::string name(instantiation ? instantiation : "");
    const bool tile = name.find("admm_tile_kernel<") != std::string::npos;
    if (!tile && name.find("admm_solve_kernel<") == std::string::npos) { if (err) *err = "not an instantiation of admm_solve_kernel / admm_tile_kernel"; return -1; }
    std::lock_guard<std::mutex> lk(g_mu);
    const Code& c = code_for(name, tile);
    if (from_disk) *from_disk = c.from_disk ? 1 : 0;
    if (c.blob.empty()) { if (err) *err = c.err; return -1; }
    return (long)c.blob.size();
}

// BUILD time: compile `instantiation` with THIS process's hipRTC (no cache is consulted) and leave it in `dir` under its prebuilt name;
// returns the code-object size, 0 if the file was already there, < 0 on failure
long jit_prebuild(const char* instantiation, const char* dir, std::string* err) {
    const std::string name(instantiation ? instantiation : "");
    const bool tile = name.find("admm_tile_kernel<") != std::string::npos;
    if (!tile && name.find("admm_solve_kernel<") == std::string::npos) { if (err) *err = "not an instantiation of admm_solve_kernel / admm_tile_kernel"; return -1; }
    const std::string path = prebuilt_path(dir ? std::string(dir) : prebuilt_dir(), name);
    if (path.empty()) { if (err) *err = "no prebuilt directory"; return -1; }
    if (err) *err = path;                            // (on success the message is the file: the build keeps exactly the files it was told about)
    Code have;
    if (disk_load(path, &have)) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    // (compile() looks the caches up first: keep it away from both for this call)
    const char* keep_cache = getenv("TINYMPC_AMD_JIT_CACHE");
    const std::string cache_val = keep_cache ? keep_cache : "";
    const char* keep_pre = getenv("TINYMPC_AMD_JIT_PREBUILT");
    const std::string pre_val = keep_pre ? keep_pre : "";
    unsetenv("TINYMPC_AMD_JIT_CACHE");
    setenv("TINYMPC_AMD_JIT_PREBUILT", "0", 1);
    Code c = compile(name, tile);
    if (keep_cache) setenv("TINYMPC_AMD_JIT_CACHE", cache_val.c_str(), 1);
    if (keep_pre) setenv("TINYMPC_AMD_JIT_PREBUILT", pre_val.c_str(), 1); else unsetenv("TINYMPC_AMD_JIT_PREBUILT");
    if (c.blob.empty()) { if (err) *err = c.err; return -1; }
    disk_store(path, c);
    if (!disk_load(path, &have)) { if (err) *err = "could not write " + path; return -1; }
    return (long)c.blob.size();
}

int jit_used_names(std::string* out) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (const auto& kv : g_code) {
        if (kv.second.blob.empty()) continue;
        if (out) { *out += kv.first; *out += '\n'; }
        ++n;
    }
    return n;
}

hipFunction_t jit_solve_kernel(const JitKey& k, std::string* err) {
    char name[256];
    snprintf(name, sizeof(name), "tinympc_amd::admm_solve_kernel<%d, %d, %d, %s, %s, %d, %d, %s, %d, %s%s>", k.nx, k.nu, k.N,
             k.soc ? "true" : "false", k.dbg ? "true" : "false", k.mode, k.lin, k.het ? "true" : "false", k.kmax, k.adapt ? "true" : "false", k.ub ? ", true" : "");
    return get(name, false, err);
}

hipFunction_t jit_tile_kernel(int nx, int nu, int N, int W, int R, int soc, int lin, int kmax, std::string* err, bool dyn, bool ub, int ext, int lm) {
    char name[256];
    // (ext: the EXT forms -- reference window / reset_duals / cold starts / store masks, per-instance problem data -- on the
    // LDS-offload set `lm` of a box form of the shape, static or dynamic; lm = 0, dyn = false: the all-in-registers form)
    if (ext) {
        snprintf(name, sizeof(name), "tinympc_amd::admm_tile_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %s, %d, %s, %d>", nx, nu, N, W, R, soc, lin, kmax, ub ? "true" : "false",
                 lm, dyn ? "true" : "false", ext);
        return get(name, true, err);
    }
    // (dyn: the dynamic slot form -- persistent grid, slots draw instances from SolveArgs::work_counter; plain variants only)
    if (dyn) snprintf(name, sizeof(name), "tinympc_amd::admm_tile_kernel<%d, %d, %d, %d, %d, %d, %d, %d, false, 0, true>", nx, nu, N, W, R, soc, lin, kmax);
    else snprintf(name, sizeof(name), "tinympc_amd::admm_tile_kernel<%d, %d, %d, %d, %d, %d, %d, %d%s>", nx, nu, N, W, R, soc, lin, kmax, ub ? ", true" : "");
    return get(name, true, err);
}

}  // namespace tinympc_amd
