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

// Coordinates of the resident Gauss points are worked out INSIDE the generated kernel (round 5): the parametric ones are three
// node tables, the physical ones the tensor-product evaluation of the control net with the geometry basis values at the Gauss
// nodes -- the tables the assembly kernels use.  (Round 4 materialised 3 x npts doubles first -- 6.3 GB at C4 -- and read them
// back; the reference fuses its inputs into the field loop the same way: pyiga/codegen/cython.py:673-701.)
// The struct is part of the generated source AND of this file: keep them identical.
struct IgxCo {
    const double *nodes[3];        // Gauss nodes of grid axis k
    const double *V[3];            // geometry basis values at those nodes: [G][P][2] (value, derivative)
    const int *fa[3];              // first active control index at node g
    int P[3], N[3];                // active functions / control points per grid axis
    const double *ctrl;            // control net (N0, N1[, N2], nc), homogeneous for NURBS
    int nc, dim, nurbs, parametric;
    int g0_lo, L1, L2, b1, b2;     // resident window of the Gauss grid (PatchDev)
};
static const char *const RTC_PRELUDE = R"IGX(
struct IgxCo {
    const double *nodes[3];
    const double *V[3];
    const int *fa[3];
    int P[3], N[3];
    const double *ctrl;
    int nc, dim, nurbs, parametric;
    int g0_lo, L1, L2, b1, b2;
};
// index of this thread's resident Gauss point and its grid indices: the launch grid is (points of the last axis, mid axis,
// axis 0) -- no integer division per point; returns -1 past the end of a line
__device__ inline long long igx_point(const IgxCo &c, int g[3])
{
    const int Ll = c.dim == 3 ? c.L2 : c.L1;
    const int gl = blockIdx.x * blockDim.x + threadIdx.x;
    if (gl >= Ll) return -1;
    g[0] = g[1] = g[2] = 0;
    if (c.dim == 3) { g[2] = gl + c.b2; g[1] = (int)blockIdx.y + c.b1; g[0] = (int)blockIdx.z + c.g0_lo; return ((long long)blockIdx.z * c.L1 + blockIdx.y) * c.L2 + gl; }
    g[1] = gl + c.b1; g[0] = (int)blockIdx.y + c.g0_lo;
    return (long long)blockIdx.y * c.L1 + gl;
}
// (x, y, z) of the Gauss point with grid indices g: x belongs to the LAST grid axis
__device__ inline void igx_coords(const IgxCo &c, const int g[3], double &x, double &y, double &z)
{
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (c.parametric) {
        for (int k = 0; k < c.dim; ++k) v[k] = c.nodes[c.dim - 1 - k][g[c.dim - 1 - k]];
    } else {
        const int P0 = c.P[0], P1 = c.P[1], P2 = c.dim == 3 ? c.P[2] : 1;
        const double *V0 = c.V[0] + (long long)g[0] * P0 * 2, *V1 = c.V[1] + (long long)g[1] * P1 * 2;
        const double *V2 = c.dim == 3 ? c.V[2] + (long long)g[2] * P2 * 2 : 0;
        const int f0 = c.fa[0][g[0]], f1 = c.fa[1][g[1]], f2 = c.dim == 3 ? c.fa[2][g[2]] : 0;
        const int N2 = c.dim == 3 ? c.N[2] : 1;
        for (int a0 = 0; a0 < P0; ++a0)
            for (int a1 = 0; a1 < P1; ++a1) {
                const double w01 = V0[2 * a0] * V1[2 * a1];
                for (int a2 = 0; a2 < P2; ++a2) {
                    const double w = c.dim == 3 ? w01 * V2[2 * a2] : w01;
                    const double *cp = c.ctrl + (((long long)(f0 + a0) * c.N[1] + (f1 + a1)) * N2 + (f2 + a2)) * c.nc;
                    for (int k = 0; k < c.nc; ++k) v[k] += w * cp[k];
                }
            }
        if (c.nurbs) {
            const double iW = 1.0 / v[c.nc - 1];
            for (int k = 0; k < c.dim; ++k) v[k] *= iW;
        }
        if (c.dim == 2) v[2] = 0.0;
    }
    x = v[0]; y = v[1]; z = c.dim == 3 ? v[2] : 0.0;
}
)IGX";

static std::string coeff_source(const char *expr)
{
    std::string s;
    s += "// generated by libigx (igx_patch_set_coeff_expr): scalar coefficient in the physical (or parametric) coordinates\n";
    s += RTC_PRELUDE;
    s += "extern \"C\" __global__ void igx_coeff_expr(const IgxCo co, double *out, long long n)\n{\n";
    s += "    int g[3];\n    const long long i = igx_point(co, g);\n    if (i < 0 || i >= n) return;\n";
    s += "    double x, y, z;\n    igx_coords(co, g, x, y, z);\n    const double pi = 3.14159265358979323846;\n    (void)x; (void)y; (void)z; (void)pi;\n";
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

// out[k][i] = expr_k(x_i, y_i, z_i) for the n_expr expressions, one kernel, coordinates evaluated in the kernel: no workspace,
// no synchronisation -- the launch is ordered on the patch's stream like every other kernel of the library
static int launch_exprs(hipStream_t st, igx_patch *pt, const std::string &src, const char *entry, double *d_out, int *hit, bool parametric = false)
{
    if (!parametric && pt->geo_kind == IGX_GEO_JACOBIAN) { set_error("a coefficient expression needs a spline geometry (physical coordinates)"); return IGX_ERR_UNSUPPORTED; }
    hipFunction_t fn;
    if (int rc = rtc_function(pt, src, entry, &fn, hit)) return rc;
    const long long n = pt->dev.npts_loc;
    if (n == 0) return IGX_OK;
    IgxCo co{};
    const int dim = pt->dim;
    for (int k = 0; k < 3; ++k) {
        co.nodes[k] = k < dim ? pt->ax[k].d_nodes : nullptr;
        co.V[k] = k < dim ? pt->gax[k].d_V : nullptr;
        co.fa[k] = k < dim ? pt->gax[k].d_fa : nullptr;
        co.P[k] = k < dim ? pt->gax[k].P : 1;
        co.N[k] = k < dim ? pt->gax[k].N : 1;
    }
    co.ctrl = pt->d_ctrl; co.nc = pt->ncomp; co.dim = dim; co.nurbs = pt->geo_kind == IGX_GEO_NURBS ? 1 : 0; co.parametric = parametric ? 1 : 0;
    co.g0_lo = pt->dev.g0_lo; co.L1 = pt->dev.L1; co.L2 = dim == 3 ? pt->dev.L2 : 1; co.b1 = pt->dev.b1; co.b2 = pt->dev.b2;
    long long nn = n;
    void *args[] = {(void *)&co, (void *)&d_out, (void *)&nn};
    // grid: (last axis in blocks of 256 points, mid axis, axis 0) -- the kernel takes its grid indices from the block indices
    const int Ll = dim == 3 ? pt->dev.L2 : pt->dev.L1;
    const unsigned gx = (unsigned)((Ll + 255) / 256), gy = (unsigned)(dim == 3 ? pt->dev.L1 : pt->dev.G0_loc), gz = (unsigned)(dim == 3 ? pt->dev.G0_loc : 1);
    if (gy > 65535u || gz > 65535u) { set_error("coefficient expression: Gauss grid beyond the launch grid"); return IGX_ERR_UNSUPPORTED; }
    if (hipModuleLaunchKernel(fn, gx, gy, gz, 256, 1, 1, 0, st, args, nullptr) != hipSuccess) { (void)hipGetLastError(); set_error("launch of the compiled coefficient kernel failed"); return IGX_ERR_HIP; }
    return IGX_OK;
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
    s += RTC_PRELUDE;
    s += "extern \"C\" __global__ void igx_form_expr(const IgxCo co, double *out, long long n)\n{\n";
    s += "    int g[3];\n    const long long i = igx_point(co, g);\n    if (i < 0 || i >= n) return;\n";
    s += "    double x, y, z;\n    igx_coords(co, g, x, y, z);\n    const double pi = 3.14159265358979323846;\n    (void)x; (void)y; (void)z; (void)pi;\n";
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
