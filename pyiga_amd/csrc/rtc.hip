// Run-time compiled coefficient kernels: the device-side counterpart of the reference's form compiler for the one thing the
// jet-form kernels cannot express themselves -- a coefficient given as an EXPRESSION in the physical coordinates.
//
// The reference generates Cython for a whole VForm, compiles it and caches the module by the hash of its source
// (pyiga/compile.py:58-73,120-132; pyiga/codegen/cython.py:325-387).  Here the forms are covered by the jet-form kernels (first
// order) and the fixed kernels; what used to be sampled on the HOST -- one double per Gauss point, 1.5 GB at C5, 0.9 s of
// numpy + PCIe -- is a scalar function c(x, y, z).  igx_patch_set_coeff_expr() takes it as a C expression, emits a kernel
//     out[i] = (expr)(X[i], Y[i], Z[i])
// compiles it with hiprtc for the device's architecture, keeps the code object ON DISK under a name made of the hash of the
// source (+ architecture), and runs it on the physical coordinates of the resident Gauss points, which the library evaluates
// from the geometry map itself (k_coeff_affine with unit coefficients).  hiprtc is loaded lazily (dlopen): a box without it can
// still use everything else; the entry point then fails loudly.
#include "igx_internal.h"
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace igx {

// ---- hiprtc through dlopen (prototypes as in hip/hiprtc.h; the handle is an opaque pointer)
typedef void *rtc_prog;
struct RtcApi {
    void *h = nullptr;
    int (*create)(rtc_prog *, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*compile)(rtc_prog, int, const char **) = nullptr;
    int (*code_size)(rtc_prog, size_t *) = nullptr;
    int (*code)(rtc_prog, char *) = nullptr;
    int (*log_size)(rtc_prog, size_t *) = nullptr;
    int (*log)(rtc_prog, char *) = nullptr;
    int (*destroy)(rtc_prog *) = nullptr;
};

static int rtc_api(RtcApi **out)
{
    static RtcApi api;
    static std::once_flag once;
    if (getenv("IGX_NO_HIPRTC")) {                           // (tests: a box without the run-time compiler)
        set_error("run-time compilation needs libhiprtc.so (switched off: IGX_NO_HIPRTC)");
        return IGX_ERR_UNSUPPORTED;
    }
    std::call_once(once, [] {
        for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6", "/opt/rocm/lib/libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so.6"}) {
            api.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.h) break;
        }
        if (!api.h) {
            // next to the HIP runtime this library is linked against (a ROCm tree outside the loader's search path)
            Dl_info di;
            if (dladdr((void *)&hipGetLastError, &di) && di.dli_fname) {
                std::string dir(di.dli_fname);
                const size_t k = dir.rfind('/');
                if (k != std::string::npos) {
                    dir.resize(k + 1);
                    for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6"}) {
                        api.h = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
                        if (api.h) break;
                    }
                }
            }
        }
        if (!api.h) return;
        api.create = (decltype(api.create))dlsym(api.h, "hiprtcCreateProgram");
        api.compile = (decltype(api.compile))dlsym(api.h, "hiprtcCompileProgram");
        api.code_size = (decltype(api.code_size))dlsym(api.h, "hiprtcGetCodeSize");
        api.code = (decltype(api.code))dlsym(api.h, "hiprtcGetCode");
        api.log_size = (decltype(api.log_size))dlsym(api.h, "hiprtcGetProgramLogSize");
        api.log = (decltype(api.log))dlsym(api.h, "hiprtcGetProgramLog");
        api.destroy = (decltype(api.destroy))dlsym(api.h, "hiprtcDestroyProgram");
    });
    if (!api.h || !api.create || !api.compile || !api.code_size || !api.code || !api.destroy) {
        set_error("run-time compilation needs libhiprtc.so (not found or incomplete)");
        return IGX_ERR_UNSUPPORTED;
    }
    *out = &api;
    return IGX_OK;
}

// ---- cache: <dir>/igx_<hash>.hsaco, dir = $IGX_CACHE_DIR | $XDG_CACHE_HOME/igx | $HOME/.cache/igx | /tmp/igx-cache-<uid>
static std::string cache_dir()
{
    std::string d;
    if (const char *e = getenv("IGX_CACHE_DIR")) d = e;
    else if (const char *x = getenv("XDG_CACHE_HOME")) d = std::string(x) + "/igx";
    else if (const char *h = getenv("HOME")) d = std::string(h) + "/.cache/igx";
    else d = "/tmp/igx-cache-" + std::to_string((long)getuid());
    std::string cur;
    for (size_t i = 0; i <= d.size(); ++i) {                 // mkdir -p
        if (i == d.size() || d[i] == '/') {
            if (!cur.empty()) (void)mkdir(cur.c_str(), 0700);
        }
        if (i < d.size()) cur.push_back(d[i]);
    }
    // the cache holds code that will be LOADED: a directory this user does not own, or that others may write to, is not used
    // (compiled in memory every time instead; rtc_code_object treats an empty directory name as "no cache")
    struct stat sb;
    if (stat(d.c_str(), &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != getuid() || (sb.st_mode & (S_IWGRP | S_IWOTH))) return std::string();
    return d;
}

// 128 bits of FNV-1a (two offset bases): a cache key, not a security boundary
static std::string source_hash(const std::string &s)
{
    unsigned long long a = 1469598103934665603ULL, b = 0x9ae16a3b2f90404fULL;
    for (unsigned char c : s) { a = (a ^ c) * 1099511628211ULL; b = (b ^ (unsigned char)(c + 0x5b)) * 0x100000001b3ULL; b ^= b >> 29; }
    char buf[40];
    snprintf(buf, sizeof buf, "%016llx%016llx", a, b);
    return buf;
}

static std::string coeff_source(const char *expr)
{
    std::string s;
    s += "// generated by libigx (igx_patch_set_coeff_expr): scalar coefficient in the physical coordinates\n";
    s += "extern \"C\" __global__ void igx_coeff_expr(const double *X, const double *Y, const double *Z, double *out, long long n)\n{\n";
    s += "    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;\n    if (i >= n) return;\n";
    s += "    const double x = X[i], y = Y[i], z = Z[i];\n    const double pi = 3.14159265358979323846;\n    (void)x; (void)y; (void)z; (void)pi;\n";
    s += "    out[i] = (double)(";
    s += expr;
    s += ");\n}\n";
    return s;
}

// code object of `src` for `arch`: from the cache, or compiled and put there.  *hit = 1 when no compilation was needed.
static int rtc_code_object(const std::string &src, const std::string &arch, std::vector<char> &code, std::string &path, int *hit)
{
    const std::string cdir = cache_dir();                   // empty: no directory this user owns alone -- nothing is read or written
    path = cdir.empty() ? std::string() : cdir + "/igx_" + source_hash(arch + "\n" + src) + ".hsaco";
    if (hit) *hit = 0;
    if (FILE *f = path.empty() ? nullptr : fopen(path.c_str(), "rb")) {
        fseek(f, 0, SEEK_END);
        const long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        code.resize(n > 0 ? (size_t)n : 0);
        const size_t got = n > 0 ? fread(code.data(), 1, (size_t)n, f) : 0;
        fclose(f);
        if (n > 4 && got == (size_t)n && memcmp(code.data(), "\x7f" "ELF", 4) == 0) { if (hit) *hit = 1; return IGX_OK; }
        code.clear();                                     // truncated or foreign file: compile again and replace it
    }
    RtcApi *api = nullptr;
    if (int rc = rtc_api(&api)) return rc;
    rtc_prog prog = nullptr;
    if (api->create(&prog, src.c_str(), "igx_coeff_expr.hip", 0, nullptr, nullptr) != 0) { set_error("hiprtcCreateProgram failed"); return IGX_ERR_HIP; }
    const std::string oarch = "--offload-arch=" + arch;
    const char *opts[] = {oarch.c_str(), "-O3", "-ffp-contract=off"};
    const int crc = api->compile(prog, 3, opts);
    if (crc != 0) {
        std::string log;
        size_t ls = 0;
        if (api->log_size && api->log && api->log_size(prog, &ls) == 0 && ls > 1) { log.resize(ls); api->log(prog, &log[0]); }
        api->destroy(&prog);
        set_error("run-time compilation of the coefficient expression failed: %.900s", log.c_str());
        return IGX_ERR_ARG;
    }
    size_t n = 0;
    if (api->code_size(prog, &n) != 0 || n == 0) { api->destroy(&prog); set_error("hiprtc returned no code"); return IGX_ERR_HIP; }
    code.resize(n);
    const int grc = api->code(prog, code.data());
    api->destroy(&prog);
    if (grc != 0) { set_error("hiprtcGetCode failed"); return IGX_ERR_HIP; }
    // atomically into the cache (another process may compile the same source at the same time)
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    if (FILE *f = path.empty() ? nullptr : fopen(tmp.c_str(), "wb")) {
        const bool ok = fwrite(code.data(), 1, n, f) == n;
        fclose(f);
        if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
    }
    return IGX_OK;
}

int rtc_compile_expr(const char *expr, const char *arch, char *path_out, int path_len, int *hit)
{
    if (!expr || !arch) { set_error("igx_rtc_compile: null argument"); return IGX_ERR_ARG; }
    std::vector<char> code;
    std::string path;
    if (int rc = rtc_code_object(coeff_source(expr), arch, code, path, hit)) return rc;
    if (path_out && path_len > 0) { strncpy(path_out, path.c_str(), (size_t)path_len - 1); path_out[path_len - 1] = 0; }
    return IGX_OK;
}

// loaded modules of this process, by code-object path (a patch that sets the same expression again does not reload it)
struct RtcModule { hipModule_t mod; hipFunction_t fn; };
static std::mutex g_mod_mutex;
static std::map<std::string, RtcModule> g_modules;

// the entry point `entry` of the code object of `src` on the patch's device (compiled or taken from the cache, loaded once)
static int rtc_function(igx_patch *pt, const std::string &src, const char *entry, hipFunction_t *fn, int *hit)
{
    hipDeviceProp_t prop;
    IGX_HIP(hipGetDeviceProperties(&prop, pt->ctx->device));
    std::vector<char> code;
    std::string path;
    if (int rc = rtc_code_object(src, prop.gcnArchName, code, path, hit)) return rc;
    std::lock_guard<std::mutex> lock(g_mod_mutex);
    const std::string key = path + "@" + std::to_string(pt->ctx->device);
    auto it = g_modules.find(key);
    if (it == g_modules.end()) {
        RtcModule m{};
        if (hipModuleLoadData(&m.mod, code.data()) != hipSuccess) { (void)hipGetLastError(); set_error("hipModuleLoadData of the compiled coefficient kernel failed"); return IGX_ERR_HIP; }
        if (hipModuleGetFunction(&m.fn, m.mod, entry) != hipSuccess) { (void)hipModuleUnload(m.mod); set_error("compiled coefficient kernel: entry point missing"); return IGX_ERR_HIP; }
        it = g_modules.emplace(key, m).first;
    }
    *fn = it->second.fn;
    return IGX_OK;
}

// parametric coordinates of the resident Gauss points: coordinate k (x = last grid axis) is the node of grid axis dim-1-k
__global__ void k_param_coords(const PatchDev pd, double *xyz)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x, n = pd.npts_loc;
    if (i >= n) return;
    long long r = i;
    int g[3] = {0, 0, 0};
    if (pd.dim == 3) { g[2] = (int)(r % pd.L2) + pd.b2; r /= pd.L2; }
    g[1] = (int)(r % pd.L1) + pd.b1; r /= pd.L1;
    g[0] = (int)r + pd.g0_lo;
    for (int k = 0; k < 3; ++k) xyz[(long long)k * n + i] = k < pd.dim ? pd.ax[pd.dim - 1 - k].nodes[g[pd.dim - 1 - k]] : 0.0;
}

// physical (or parametric) coordinates of the resident Gauss points, [3][npts_loc] (absent axes: zeros): the affine-coefficient
// kernel with unit coefficients.  The caller frees *xyz.
static int physical_coordinates(hipStream_t st, igx_patch *pt, double **xyz, bool parametric = false)
{
    const long long n = pt->dev.npts_loc;
    if (parametric) {
        *xyz = nullptr;
        if (hipMalloc((void **)xyz, (size_t)3 * std::max<long long>(n, 1) * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); set_error("coefficient expression: %.2f GB for the coordinates", 24e-9 * n); return IGX_ERR_NOMEM; }
        if (n) k_param_coords<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(pt->dev, *xyz);
        if (hipGetLastError() != hipSuccess) { (void)hipFree(*xyz); *xyz = nullptr; set_error("parametric coordinates: launch failed"); return IGX_ERR_HIP; }
        return IGX_OK;
    }
    *xyz = nullptr;
    if (hipMalloc((void **)xyz, (size_t)3 * std::max<long long>(n, 1) * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); set_error("coefficient expression: %.2f GB for the coordinates", 24e-9 * n); return IGX_ERR_NOMEM; }
    int rc = IGX_OK;
    for (int k = 0; k < 3 && !rc; ++k) {
        double c[4] = {0, 0, 0, 0};
        if (k < pt->dim) c[1 + k] = 1.0;
        rc = launch_coeff_affine(st, pt, c, *xyz + (size_t)k * n);
    }
    if (rc) { (void)hipFree(*xyz); *xyz = nullptr; }
    return rc;
}

// out[k][i] = expr_k(X[i], Y[i], Z[i]) for the n_expr expressions, one kernel
static int launch_exprs(hipStream_t st, igx_patch *pt, const std::string &src, const char *entry, double *d_out, int *hit, bool parametric = false)
{
    if (!parametric && pt->geo_kind == IGX_GEO_JACOBIAN) { set_error("a coefficient expression needs a spline geometry (physical coordinates)"); return IGX_ERR_UNSUPPORTED; }
    hipFunction_t fn;
    if (int rc = rtc_function(pt, src, entry, &fn, hit)) return rc;
    const long long n = pt->dev.npts_loc;
    if (n == 0) return IGX_OK;
    double *xyz = nullptr;
    int rc = physical_coordinates(st, pt, &xyz, parametric);
    if (rc) return rc;
    {
        const double *X = xyz, *Y = xyz + n, *Z = xyz + 2 * n;
        long long nn = n;
        void *args[] = {(void *)&X, (void *)&Y, (void *)&Z, (void *)&d_out, (void *)&nn};
        const unsigned grid = (unsigned)((n + 255) / 256);
        if (hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, st, args, nullptr) != hipSuccess) { (void)hipGetLastError(); set_error("launch of the compiled coefficient kernel failed"); rc = IGX_ERR_HIP; }
    }
    const hipError_t se = hipStreamSynchronize(st);
    (void)hipFree(xyz);
    if (!rc && se != hipSuccess) { set_error("compiled coefficient kernel: %s", hipGetErrorString(se)); rc = IGX_ERR_HIP; }
    return rc;
}

int launch_coeff_expr(hipStream_t st, igx_patch *pt, const char *expr, double *d_coeff, int *hit, bool parametric)
{
    return launch_exprs(st, pt, coeff_source(expr), "igx_coeff_expr", d_coeff, hit, parametric);
}

// The coefficient fields of a whole form in ONE generated kernel: out[k * n + i] = expr_k at point i.
static std::string form_source(int n_expr, const char *const *expr)
{
    std::string s;
    s += "// generated by libigx (igx_patch_set_form_expr): coefficient fields of a form in the physical coordinates\n";
    s += "extern \"C\" __global__ void igx_form_expr(const double *X, const double *Y, const double *Z, double *out, long long n)\n{\n";
    s += "    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;\n    if (i >= n) return;\n";
    s += "    const double x = X[i], y = Y[i], z = Z[i];\n    const double pi = 3.14159265358979323846;\n    (void)x; (void)y; (void)z; (void)pi;\n";
    for (int k = 0; k < n_expr; ++k) {
        s += "    out[" + std::to_string(k) + " * n + i] = (double)(";
        s += expr[k];
        s += ");\n";
    }
    s += "}\n";
    return s;
}

int launch_form_exprs(hipStream_t st, igx_patch *pt, int n_expr, const char *const *expr, double *d_out, int *hit)
{
    return launch_exprs(st, pt, form_source(n_expr, expr), "igx_form_expr", d_out, hit);
}

int rtc_compile_form(int n_expr, const char *const *expr, const char *arch, char *path_out, int path_len, int *hit)
{
    if (!expr || !arch || n_expr < 1) { set_error("igx_rtc_compile_form: bad argument"); return IGX_ERR_ARG; }
    std::vector<char> code;
    std::string path;
    if (int rc = rtc_code_object(form_source(n_expr, expr), arch, code, path, hit)) return rc;
    if (path_out && path_len > 0) { strncpy(path_out, path.c_str(), (size_t)path_len - 1); path_out[path_len - 1] = 0; }
    return IGX_OK;
}

} // namespace igx
