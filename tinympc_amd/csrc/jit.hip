// jit.hip -- see jit.hpp.
#include "jit.hpp"

#include <dlfcn.h>

#include <cstdio>
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
    r.ok = r.create && r.add_name && r.compile && r.log_size && r.log && r.lowered && r.code_size && r.code && r.destroy;
    return r;
}

struct Entry { hipFunction_t fn = nullptr; std::string err; };
std::map<std::string, Entry> g_cache;                   // by instantiation name
std::mutex g_mu;

Entry build(const std::string& name_s, const bool tile) {
    Entry e;
    const char* name = name_s.c_str();
    Rtc& R = rtc();
    if (!R.ok) { e.err = "libhiprtc is not available"; return e; }
    // hipRTC brings its own runtime header: the two system includes of the kernel header are dropped
    std::string hdr(kAdmmKernelSrc);
    for (const char* inc : {"#include <hip/hip_runtime.h>", "#include <stdint.h>"}) {
        const size_t p = hdr.find(inc);
        if (p != std::string::npos) hdr.replace(p, std::string(inc).size(), "");
    }
    const std::string src = tile ? "#include \"tile_kernel.hip.h\"\n" : "#include \"admm_kernel.hip.h\"\n";
    const char* hn[] = {"admm_kernel.hip.h", "tile_kernel.hip.h"};
    const char* hs[] = {hdr.c_str(), kTileKernelSrc};
    Rtc::Program prog = nullptr;
    if (R.create(&prog, src.c_str(), "tinympc_amd_jit.hip", 2, hs, hn) != 0) { e.err = "hiprtcCreateProgram failed"; return e; }
    R.add_name(prog, name);
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    const int rc = R.compile(prog, 3, opts);
    if (rc != 0) {
        size_t n = 0;
        R.log_size(prog, &n);
        std::string log(n, '\0');
        if (n) R.log(prog, &log[0]);
        e.err = "hipRTC compilation of " + std::string(name) + " failed: " + log.substr(0, 400);
        R.destroy(&prog);
        return e;
    }
    const char* low = nullptr;
    size_t cs = 0;
    if (R.lowered(prog, name, &low) != 0 || R.code_size(prog, &cs) != 0 || cs == 0) { e.err = "hipRTC produced no code"; R.destroy(&prog); return e; }
    std::vector<char> code(cs);
    R.code(prog, code.data());
    const std::string lowered(low);
    R.destroy(&prog);
    hipModule_t mod = nullptr;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) { (void)hipGetLastError(); e.err = "hipModuleLoadData failed"; return e; }
    if (hipModuleGetFunction(&e.fn, mod, lowered.c_str()) != hipSuccess) { (void)hipGetLastError(); e.fn = nullptr; e.err = "kernel symbol not found in the compiled module"; }
    return e;                                            // the module stays loaded for the life of the process
}

}  // namespace

static hipFunction_t get(const std::string& name, const bool tile, std::string* err) {
    std::lock_guard<std::mutex> lk(g_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);                            // a module belongs to the device it was loaded on
    const std::string key = name + "@" + std::to_string(dev);
    auto it = g_cache.find(key);
    if (it == g_cache.end()) it = g_cache.emplace(key, build(name, tile)).first;
    if (!it->second.fn && err) *err = it->second.err;
    return it->second.fn;
}

hipFunction_t jit_solve_kernel(const JitKey& k, std::string* err) {
    char name[256];
    snprintf(name, sizeof(name), "tinympc_amd::admm_solve_kernel<%d, %d, %d, %s, %s, %d, %d, %s, %d>", k.nx, k.nu, k.N,
             k.soc ? "true" : "false", k.dbg ? "true" : "false", k.mode, k.lin, k.het ? "true" : "false", k.kmax);
    return get(name, false, err);
}

hipFunction_t jit_tile_kernel(int nx, int nu, int N, int W, int R, bool soc, int lin, int kmax, std::string* err) {
    char name[256];
    snprintf(name, sizeof(name), "tinympc_amd::admm_tile_kernel<%d, %d, %d, %d, %d, %s, %d, %d>", nx, nu, N, W, R, soc ? "true" : "false", lin, kmax);
    return get(name, true, err);
}

}  // namespace tinympc_amd
