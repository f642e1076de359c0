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
        return IGX_ERR_NORTC;
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
        return IGX_ERR_NORTC;
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

static bool expr_is_plain(const char *expr);
static int exprs_ok(int n, const char *const *expr, const char *what)
{
    for (int k = 0; k < n; ++k)
        if (expr[k] && !expr_is_plain(expr[k])) {
            set_error("%s: the expression may not contain line breaks, backslashes, comments, '#', ';' or braces", what);
            return IGX_ERR_ARG;
        }
    return IGX_OK;
}

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
        return IGX_ERR_COMPILE;
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
    if (int rc = exprs_ok(1, &expr, "igx_rtc_compile")) return rc;
    if (int rc = rtc_code_object(coeff_source(expr), arch, code, path, hit)) return rc;
    if (path_out && path_len > 0) { strncpy(path_out, path.c_str(), (size_t)path_len - 1); path_out[path_len - 1] = 0; }
    return IGX_OK;
}

// loaded modules of this process, by code-object path (a patch that sets the same expression again does not reload it)
struct RtcModule { hipModule_t mod; hipFunction_t fn; };
static std::mutex g_mod_mutex;
static std::map<std::string, RtcModule> g_modules;

// the entry point `entry` of the code object of `src` on the patch's device: compiled or taken from the cache and loaded ONCE per
// process -- later requests for the same source (same cache file; without a cache directory: same source hash) find the loaded
// module without touching the disk
static int rtc_function(igx_patch *pt, const std::string &src, const char *entry, hipFunction_t *fn, int *hit)
{
    hipDeviceProp_t prop;
    IGX_HIP(hipGetDeviceProperties(&prop, pt->ctx->device));
    const std::string arch = prop.gcnArchName;
    const std::string cdir = cache_dir(), hash = source_hash(arch + "\n" + src);
    const std::string key = (cdir.empty() ? "mem:" + hash : cdir + "/igx_" + hash + ".hsaco") + "@" + std::to_string(pt->ctx->device) + ":" + entry;
    {
        std::lock_guard<std::mutex> lock(g_mod_mutex);
        auto it = g_modules.find(key);
        if (it != g_modules.end()) { *fn = it->second.fn; if (hit) *hit = 1; return IGX_OK; }
    }
    std::vector<char> code;
    std::string path;
    if (int rc = rtc_code_object(src, arch, code, path, hit)) return rc;
    std::lock_guard<std::mutex> lock(g_mod_mutex);
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

static void fill_co(const igx_patch *pt, bool parametric, IgxCo &co);

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
    fill_co(pt, parametric, co);
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
    if (int rc = exprs_ok(1, &expr, "coefficient expression")) return rc;
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
    if (int rc = exprs_ok(n_expr, expr, "form coefficients")) return rc;
    return launch_exprs(st, pt, form_source(n_expr, expr), "igx_form_expr", d_out, hit);
}

// ---------------------------------------------------------------------------------------------
// The FIELD kernel of a form whose physical coefficients are C expressions: geometry map, Jacobian, the coefficient
// expressions and the transformation to parametric jet coefficients (geo_device.h: fields_form) in ONE generated kernel, so
// that no coefficient ever exists as a full-grid array (before: expressions -> n arrays -> copy -> field kernel reads them
// back).  The reference fuses its inputs into the field loop of the generated assembler the same way
// (pyiga/codegen/cython.py:673-701, generate_precomp).  The geometry is evaluated line-wise like k_geo_fields_lines
// (kern_basis.hip): the control net contracted with the basis of the outer axes once per grid line into LDS, then
// (p + 1) * nc products per point.  Degree, components, the terms of the form and which coefficients exist are constants of
// the generated source: the compiler drops every product with an absent coefficient.
// The struct is part of the generated source AND of this file: keep them identical.
struct IgxFF {
    const double *V[3];            // geometry basis values at the Gauss nodes of grid axis k: [G][P][2]
    const int *fa[3];              // first active control index at node g
    int P[3], N[3];
    const double *ctrl;            // control net (N0, N1[, N2], nc), homogeneous for NURBS
    const double *w[3];            // Gauss weights per grid axis
    const double *nodes[3];        // Gauss nodes per grid axis (coefficients in the parametric coordinates)
    int g0_lo, G0loc, G1, G2, LPB, pad;
    double *fields;                // [terms][resident points]
};
static const char *const RTC_FIELDS_BODY = R"IGX(
struct IgxFF {
    const double *V[3];
    const int *fa[3];
    int P[3], N[3];
    const double *ctrl;
    const double *w[3];
    const double *nodes[3];
    int g0_lo, G0loc, G1, G2, LPB, pad;
    double *fields;
};
extern "C" __global__ void __launch_bounds__(256) igx_form_fields(const IgxFF A)
{
    extern __shared__ double Lc[];                        // [LPB][N of the last axis][nc][DIM]
    constexpr int DIM = IGX_DIM, nc = IGX_NC;
    const int LN = DIM == 3 ? A.G2 : A.G1;
    const long long nlines = DIM == 3 ? (long long)A.G0loc * A.G1 : A.G0loc;
    const long long total = nlines * LN;
    const int NgL = A.N[DIM - 1];
    const long long line0 = (long long)blockIdx.x * A.LPB;
    const int per_line = NgL * nc;
    // ---- line coefficients: value and derivatives along the outer axes of the net contracted with their basis
    for (int w = threadIdx.x; w < A.LPB * per_line; w += blockDim.x) {
        const int ll = w / per_line, rem = w - ll * per_line;
        const int cL = rem / nc, c = rem - cL * nc;
        const long long line = line0 + ll;
        if (line >= nlines) continue;
        double sv = 0.0, s0 = 0.0, s1 = 0.0;
        if (DIM == 2) {
            const int g0 = A.g0_lo + (int)line;
            const double *V0 = A.V[0] + (size_t)g0 * A.P[0] * 2;
            const int f0 = A.fa[0][g0];
            for (int a0 = 0; a0 < A.P[0]; ++a0) {
                const double cf = A.ctrl[((size_t)(f0 + a0) * A.N[1] + cL) * nc + c];
                sv += V0[a0 * 2] * cf;
                s0 += V0[a0 * 2 + 1] * cf;
            }
        } else {
            const int g0 = A.g0_lo + (int)(line / A.G1), g1 = (int)(line % A.G1);
            const double *V0 = A.V[0] + (size_t)g0 * A.P[0] * 2;
            const double *V1 = A.V[1] + (size_t)g1 * A.P[1] * 2;
            const int f0 = A.fa[0][g0], f1 = A.fa[1][g1];
            for (int a0 = 0; a0 < A.P[0]; ++a0) {
                double tv = 0.0, t1 = 0.0;
                for (int a1 = 0; a1 < A.P[1]; ++a1) {
                    const double cf = A.ctrl[(((size_t)(f0 + a0) * A.N[1] + (f1 + a1)) * A.N[2] + cL) * nc + c];
                    tv += V1[a1 * 2] * cf;
                    t1 += V1[a1 * 2 + 1] * cf;
                }
                sv += V0[a0 * 2] * tv;
                s0 += V0[a0 * 2 + 1] * tv;
                s1 += V0[a0 * 2] * t1;
            }
        }
        double *dst = Lc + (size_t)w * DIM;
        dst[0] = sv; dst[1] = s0;
        if (DIM == 3) dst[2] = s1;
    }
    __syncthreads();
    // ---- points of the line(s)
    const int PL = A.P[DIM - 1];
    const int npts_blk = A.LPB * LN;
    for (int t = threadIdx.x; t < npts_blk; t += blockDim.x) {
        const int ll = t / LN, gL = t - ll * LN;
        const long long line = line0 + ll;
        if (line >= nlines) break;
        const double *VL = A.V[DIM - 1] + (size_t)gL * PL * 2;
        const int fL = A.fa[DIM - 1][gL];
        double val[nc], jac[nc][3];
#pragma unroll
        for (int c = 0; c < nc; ++c) { val[c] = 0.0; jac[c][0] = jac[c][1] = jac[c][2] = 0.0; }
        const double *lc = Lc + ((size_t)ll * NgL + fL) * nc * DIM;
        for (int aL = 0; aL < PL; ++aL) {
            const double n = VL[aL * 2], d = VL[aL * 2 + 1];
#pragma unroll
            for (int c = 0; c < nc; ++c) {
                const double *e = lc + ((size_t)aL * nc + c) * DIM;
                val[c] += n * e[0];
                jac[c][0] += n * e[1];
                if (DIM == 3) jac[c][1] += n * e[2];
                jac[c][DIM - 1] += d * e[0];
            }
        }
        // physical point and Jacobian J[r][c] = d G_r / d xi_c, c = 0 the LAST grid axis; NURBS: quotient rule, one reciprocal
        double J[3][3], ev[3] = {0.0, 0.0, 0.0};
        if (IGX_NURBS) {
            const double Wh = val[nc - 1];
            const double iW = 1.0 / Wh, iW2 = iW * iW;
#pragma unroll
            for (int r = 0; r < DIM; ++r) {
                ev[r] = val[r] * iW;
#pragma unroll
                for (int c = 0; c < DIM; ++c) J[r][c] = (jac[r][DIM - 1 - c] * Wh - val[r] * jac[nc - 1][DIM - 1 - c]) * iW2;
            }
        } else {
#pragma unroll
            for (int r = 0; r < DIM; ++r) {
                ev[r] = val[r];
#pragma unroll
                for (int c = 0; c < DIM; ++c) J[r][c] = jac[r][DIM - 1 - c];
            }
        }
        int g0, g1;
        if (DIM == 3) { g0 = A.g0_lo + (int)(line / A.G1); g1 = (int)(line % A.G1); }
        else { g0 = A.g0_lo + (int)line; g1 = gL; }
        double GW = A.w[0][g0] * A.w[1][g1];
        if (DIM == 3) GW = GW * A.w[2][gL];
        // T = diag(1, JacInv), JacInv[a][r] = d xi_a / d x_r
        double T[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) T[r][q] = 0.0;
        T[0][0] = 1.0;
        double det;
        if (DIM == 2) {
            det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
            const double inv = 1.0 / det;
            T[1][1] = inv * J[1][1]; T[1][2] = inv * -J[0][1];
            T[2][1] = inv * -J[1][0]; T[2][2] = inv * J[0][0];
        } else {
            const double t3 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
            const double t4 = J[1][0] * J[2][2] - J[1][2] * J[2][0];
            const double t5 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
            det = (J[0][0] * t3 - J[0][1] * t4) + J[0][2] * t5;
            const double inv = 1.0 / det;
            T[1][1] = inv * t3;
            T[1][2] = inv * -(J[0][1] * J[2][2] - J[0][2] * J[2][1]);
            T[1][3] = inv * (J[0][1] * J[1][2] - J[0][2] * J[1][1]);
            T[2][1] = inv * -t4;
            T[2][2] = inv * (J[0][0] * J[2][2] - J[0][2] * J[2][0]);
            T[2][3] = inv * -(J[0][0] * J[1][2] - J[0][2] * J[1][0]);
            T[3][1] = inv * t5;
            T[3][2] = inv * -(J[0][0] * J[2][1] - J[0][1] * J[2][0]);
            T[3][3] = inv * (J[0][0] * J[1][1] - J[0][1] * J[1][0]);
        }
        const double W = GW * fabs(det);
        // the coefficients' coordinates: the physical point, or the parametric one (x belongs to the LAST grid axis)
        const double x = IGX_PARAMETRIC ? A.nodes[DIM - 1][gL] : ev[0];
        const double y = IGX_PARAMETRIC ? A.nodes[DIM - 2][DIM == 3 ? g1 : g0] : ev[1];
        const double z = DIM == 3 ? (IGX_PARAMETRIC ? A.nodes[0][g0] : ev[2]) : 0.0;
        const double pi = 3.14159265358979323846;
        (void)x; (void)y; (void)z; (void)pi;
        constexpr int NJ = DIM + 1;
        double P[4][4];
        igx_form_coefficients(P, x, y, z, pi);
        double TP[4][4];                                   // T P: rows of T act on the test index
#pragma unroll
        for (int a = 0; a < NJ; ++a)
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                double v = 0.0;
#pragma unroll
                for (int r = 0; r < NJ; ++r) v = fma(T[a][r], P[r][q], v);
                TP[a][q] = v;
            }
        const long long idx = line * LN + gL;
        // field k = W (T P T^t)[a_k][b_k] for the terms (k, a, b) of the form
#define IGX_ONE_TERM(k, a, b) { double v = 0.0; for (int q = 0; q < NJ; ++q) v = fma(TP[a][q], T[b][q], v); A.fields[(long long)(k) * total + idx] = W * v; }
        IGX_TERMS(IGX_ONE_TERM)
    }
}
)IGX";

static std::string form_fields_source(int dim, int nc, bool nurbs, const char *const expr[16], int nterms, const int *form_ab, bool parametric = false)
{
    std::string s;
    s += "// generated by libigx (igx_patch_set_form_expr): field kernel of a form -- geometry, coefficient expressions and the\n";
    s += "// transformation to parametric jet coefficients in one pass over the resident Gauss points\n";
    s += "#define IGX_DIM " + std::to_string(dim) + "\n#define IGX_NC " + std::to_string(nc) + "\n#define IGX_NURBS " + std::to_string(nurbs ? 1 : 0) + "\n";
    s += "#define IGX_PARAMETRIC " + std::to_string(parametric ? 1 : 0) + "\n";
    s += "#define IGX_TERMS(X)";
    for (int k = 0; k < nterms; ++k) s += " X(" + std::to_string(k) + ", " + std::to_string(form_ab[k] >> 2) + ", " + std::to_string(form_ab[k] & 3) + ")";
    s += "\n";
    s += "__device__ inline void igx_form_coefficients(double (&P)[4][4], const double x, const double y, const double z, const double pi)\n{\n";
    s += "    (void)x; (void)y; (void)z; (void)pi;\n";
    for (int r = 0; r < 4; ++r)
        for (int q = 0; q < 4; ++q) {
            s += "    P[" + std::to_string(r) + "][" + std::to_string(q) + "] = ";
            if (expr[4 * r + q]) { s += "(double)("; s += expr[4 * r + q]; s += ");\n"; }
            else s += "0.0;\n";
        }
    s += "}\n";
    s += RTC_FIELDS_BODY;
    return s;
}

// compile (or fetch) the field kernel of the form `expr` for the patch; the function handle stays valid for the process
int rtc_form_fields_function(igx_patch *pt, const char *const expr[16], int nterms, const int *form_ab, void **fn_out, int *hit, bool parametric)
{
    if (pt->geo_kind == IGX_GEO_JACOBIAN) { set_error("a coefficient expression needs a spline geometry (physical coordinates)"); return IGX_ERR_UNSUPPORTED; }
    if (pt->boxed) { set_error("a form given as expressions needs the whole Gauss grid of the slab (no span box)"); return IGX_ERR_UNSUPPORTED; }
    hipFunction_t fn;
    if (int rc = exprs_ok(16, expr, "form given as expressions")) return rc;
    const std::string src = form_fields_source(pt->dim, pt->ncomp, pt->geo_kind == IGX_GEO_NURBS, expr, nterms, form_ab, parametric);
    if (int rc = rtc_function(pt, src, "igx_form_fields", &fn, hit)) return rc;
    *fn_out = (void *)fn;
    return IGX_OK;
}

// terms (a, b) of the parametric form of a coefficient table: the Jacobian mixes the derivative directions of each jet block
int form_terms(int dim, const char *const expr[16], int form_ab[16])
{
    const int nj = dim + 1;
    bool blk[2][2] = {{false, false}, {false, false}};
    for (int r = 0; r < 4; ++r)
        for (int s = 0; s < 4; ++s)
            if (expr[4 * r + s]) blk[r > 0][s > 0] = true;
    int nt = 0;
    for (int a = 0; a < nj; ++a)
        for (int b = 0; b < nj; ++b)
            if (blk[a > 0][b > 0]) form_ab[nt++] = 4 * a + b;
    for (int k = nt; k < 16; ++k) form_ab[k] = 0;
    return nt;
}

int rtc_compile_form_fields(int dim, int ncomp, const char *const expr[16], const char *arch, char *path_out, int path_len, int *hit)
{
    if (!expr || !arch || dim < 2 || dim > 3 || (ncomp != dim && ncomp != dim + 1)) { set_error("igx_rtc_compile_form_fields: bad argument"); return IGX_ERR_ARG; }
    int form_ab[16];
    const int nt = form_terms(dim, expr, form_ab);
    if (nt == 0) { set_error("igx_rtc_compile_form_fields: all coefficients are absent"); return IGX_ERR_ARG; }
    std::vector<char> code;
    std::string path;
    if (int rc = exprs_ok(16, expr, "igx_rtc_compile_form_fields")) return rc;
    if (int rc = rtc_code_object(form_fields_source(dim, ncomp, ncomp == dim + 1, expr, nt, form_ab), arch, code, path, hit)) return rc;
    if (path_out && path_len > 0) { strncpy(path_out, path.c_str(), (size_t)path_len - 1); path_out[path_len - 1] = 0; }
    return IGX_OK;
}

// launch shape of k_geo_fields_lines (kern_basis.hip): lines per block and their coefficients in LDS
static void form_fields_shape(const igx_patch *pt, int *LPB_out, size_t *lds_out)
{
    const int dim = pt->dim;
    const int LN = dim == 3 ? pt->dev.L2 : pt->dev.L1;
    int LPB = std::max(1, 256 / std::max(LN, 1));
    if (LN > 256)
        for (int l = 1; l <= 8; ++l)
            if ((l * LN) % 256 == 0) { LPB = l; break; }
    *LPB_out = LPB;
    *lds_out = (size_t)LPB * pt->gax[dim - 1].N * pt->ncomp * dim * sizeof(double);
}

// can the generated field kernel serve this patch?  (a spline geometry whose control lines fit LDS, the whole grid of the slab)
bool form_fields_applicable(const igx_patch *pt)
{
    if (pt->geo_kind == IGX_GEO_JACOBIAN || pt->boxed) return false;
    int LPB;
    size_t lds;
    form_fields_shape(pt, &LPB, &lds);
    return lds <= 64 * 1024;
}

int launch_form_fields(hipStream_t st, const igx_patch *pt, void *fn, double *d_fields)
{
    const int dim = pt->dim;
    const PatchDev &pd = pt->dev;
    const long long total = pd.npts_loc;
    if (total == 0) return IGX_OK;
    const int G1 = pd.L1, G2 = dim == 3 ? pd.L2 : 1, LN = dim == 3 ? G2 : G1;
    int LPB;
    size_t lds;
    form_fields_shape(pt, &LPB, &lds);
    if (lds > 64 * 1024) { set_error("field kernel of a form given as expressions: the geometry's control lines do not fit LDS (%zu bytes)", lds); return IGX_ERR_UNSUPPORTED; }
    IgxFF a{};
    for (int k = 0; k < 3; ++k) {
        a.V[k] = k < dim ? pt->gax[k].d_V : nullptr;
        a.fa[k] = k < dim ? pt->gax[k].d_fa : nullptr;
        a.P[k] = k < dim ? pt->gax[k].P : 1;
        a.N[k] = k < dim ? pt->gax[k].N : 1;
        a.w[k] = k < dim ? pd.ax[k].w : nullptr;
        a.nodes[k] = k < dim ? pt->ax[k].d_nodes : nullptr;
    }
    a.ctrl = pt->d_ctrl;
    a.g0_lo = pd.g0_lo; a.G0loc = pd.G0_loc; a.G1 = G1; a.G2 = G2; a.LPB = LPB; a.fields = d_fields;
    const long long nlines = total / LN;
    void *args[] = {(void *)&a};
    if (hipModuleLaunchKernel((hipFunction_t)fn, (unsigned)((nlines + LPB - 1) / LPB), 1, 1, 256, 1, 1, (unsigned)lds, st, args, nullptr) != hipSuccess) {
        (void)hipGetLastError(); set_error("launch of the compiled field kernel failed"); return IGX_ERR_HIP;
    }
    return IGX_OK;
}

// ---------------------------------------------------------------------------------------------
// The first two contractions of the 3D load vector (kern_vector.hip, k_lv12) with the FUNCTION inside: the generated kernel
// evaluates f at the points of its grid line (parametric coordinates from the node tables, physical ones from the geometry
// map) and multiplies with the resident weight field -- the function values never exist as an array (before: a generated
// kernel wrote 8 bytes per Gauss point, k_lv12 read them back).  Same walk, same chunks, same summation order as k_lv12.
// AxisDev is part of the generated source AND of igx_internal.h: keep them identical.
static const char *const RTC_LV12_BODY = R"IGX(
struct AxisDev {
    int p, P, N, n, q, G;
    int S;
    const double *nodes;
    const double *w;
    const double *V;
    const double *PI;
    const int *fa;
    const int *mslo, *mshi;
    const int *jlo, *jhi;
    const int *rp;
    const int *pair_i, *pair_j;
};
__device__ inline double igx_f_at(const IgxCo &co, const int g0, const int g1, const int g2)
{
    double x, y, z;
    if (IGX_PARAMETRIC) { x = co.nodes[2][g2]; y = co.nodes[1][g1]; z = co.nodes[0][g0]; }
    else { const int g[3] = {g0, g1, g2}; igx_coords(co, g, x, y, z); }
    const double pi = 3.14159265358979323846;
    (void)pi;
    return (double)(IGX_F);
}
extern "C" __global__ void __launch_bounds__(256) igx_lv12_expr(const IgxCo co, const double *__restrict__ wfield, double *__restrict__ t2,
                                                                const AxisDev a1, const AxisDev a2, int G0, int chunk_spans, int nchunks)
{
    constexpr int P = IGX_P, MAXPC = 5, NPASS = IGX_NPASS, LV_WAVES = 4;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int q = a2.q, PQ = P * q, N2 = a2.N, G2 = a2.G;
    double *Vt = lds;                                    // [PQ][N2]: basis value of dof i2 at point k of its support (0 past its end)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *buf = lds + ((PQ * N2 + 1) & ~1) + wave * ((G2 + 1) & ~1);    // this wave's line of products
    for (int e = threadIdx.x; e < PQ * N2; e += blockDim.x) {
        const int k = e / N2, i = e - k * N2;
        const int s_lo = a2.mslo[i], nsup = (a2.mshi[i] - s_lo) * q;
        double v = 0.0;
        if (k < nsup) { const int g = s_lo * q + k; v = a2.V[((size_t)g * P + (i - a2.fa[g / q])) * 2]; }
        Vt[e] = v;
    }
    __syncthreads();
    const long long unit = (long long)blockIdx.x * LV_WAVES + wave;
    if (unit >= (long long)G0 * nchunks) return;
    const int g0 = (int)(unit / nchunks), ch = (int)(unit - (long long)g0 * nchunks);
    const int s_a = ch * chunk_spans, s_b = ch == nchunks - 1 ? a1.n : s_a + chunk_spans;
    const int q1 = a1.q, G1 = a1.G, N1 = a1.N;
    int gfirst[NPASS];
#pragma unroll
    for (int k = 0; k < NPASS; ++k) { const int i2 = min(lane + 64 * k, N2 - 1); gfirst[k] = a2.mslo[i2] * q; }
    double acc[P][NPASS];
#pragma unroll
    for (int a = 0; a < P; ++a)
#pragma unroll
        for (int k = 0; k < NPASS; ++k) acc[a][k] = 0.0;
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int npc = (G2 / 2 + 63) >> 6;
    d2 vw[MAXPC];
    auto request = [&](const int g1) {
        const d2 *pw = (const d2 *)(wfield + ((long long)g0 * G1 + g1) * G2);
#pragma unroll
        for (int c = 0; c < MAXPC; ++c)
            if (c < npc) vw[c] = pw[min(lane + 64 * c, G2 / 2 - 1)];
    };
    const int g_a = s_a * q1, g_b = s_b * q1;
    request(g_a);
    int l = 0, sp = s_a;
    for (int g1 = g_a; g1 < g_b; ++g1) {
        // products W f of this line -> LDS; then the loads of the next line
#pragma unroll
        for (int c = 0; c < MAXPC; ++c)
            if (c < npc) {
                const int e = lane + 64 * c;
                if (e < G2 / 2) {
                    d2 x;
                    x.x = igx_f_at(co, g0 + co.g0_lo, g1, 2 * e) * vw[c].x;
                    x.y = igx_f_at(co, g0 + co.g0_lo, g1, 2 * e + 1) * vw[c].y;
                    ((d2 *)buf)[e] = x;
                }
            }
        if (g1 + 1 < g_b) request(g1 + 1);
        double v1[P];
#pragma unroll
        for (int a = 0; a < P; ++a) v1[a] = a1.V[((size_t)g1 * P + a) * 2];
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            const int i2 = min(lane + 64 * k, N2 - 1);
            const double *bl = buf + gfirst[k];
            double r = 0.0;
            for (int m = 0; m < PQ; ++m) r = fma(Vt[m * N2 + i2], bl[min(m, G2 - 1 - gfirst[k])], r);
#pragma unroll
            for (int a = 0; a < P; ++a) acc[a][k] = fma(v1[a], r, acc[a][k]);
        }
        if (++l < q1) continue;
        // end of span sp: the dofs that leave the active set; at the end of the chunk every dof that is still active
        const int base = a1.fa[sp];
        const int nleave = sp + 1 < a1.n ? (sp + 1 < s_b ? a1.fa[sp + 1] - base : P) : P;
        for (int j = 0; j < nleave; ++j) {
            const int i1 = base + j;
            if (i1 < N1) {
                const bool whole = a1.mslo[i1] >= s_a && (sp + 1 < s_b || sp + 1 == a1.n || j < a1.fa[min(sp + 1, a1.n - 1)] - base);
                double *dst = t2 + ((long long)g0 * N1 + i1) * N2;
#pragma unroll
                for (int k = 0; k < NPASS; ++k)
                    if (lane + 64 * k < N2) {
                        if (whole) dst[lane + 64 * k] = acc[0][k];
                        else (void)__hip_atomic_fetch_add(dst + lane + 64 * k, acc[0][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
            }
#pragma unroll
            for (int a = 0; a < P - 1; ++a)
#pragma unroll
                for (int k = 0; k < NPASS; ++k) acc[a][k] = acc[a + 1][k];
#pragma unroll
            for (int k = 0; k < NPASS; ++k) acc[P - 1][k] = 0.0;
        }
        l = 0; ++sp;
    }
}
)IGX";

// An expression is pasted into generated source: one that could end the construct it is pasted into -- a line break, a line
// continuation, a comment opener, a preprocessor line -- is refused before anything is generated (expressions traced in Python are
// single-line arithmetic; this guards the public C entry points).
static bool expr_is_plain(const char *expr)
{
    for (const char *c = expr; *c; ++c) {
        if (*c == '\n' || *c == '\r' || *c == '\\' || *c == '#' || *c == ';' || *c == '{' || *c == '}') return false;
        if (c[0] == '/' && (c[1] == '/' || c[1] == '*')) return false;
    }
    return true;
}

static std::string lv12_source(int P, int npass, bool parametric, const char *expr)
{
    std::string s;
    s += "// generated by libigx (igx_load_vector_expr): first two contractions of the 3D load vector with the function inside\n";
    s += "#define IGX_P " + std::to_string(P) + "\n#define IGX_NPASS " + std::to_string(npass) + "\n#define IGX_PARAMETRIC " + std::to_string(parametric ? 1 : 0) + "\n";
    s += RTC_PRELUDE;
    // (the function as a FUNCTION behind the prelude, like the coefficient kernels: not a macro in front of it, whose body the
    // names of the prelude could capture)
    s += "__device__ inline double igx_f(const double x, const double y, const double z, const double pi)\n{\n    return (double)(";
    s += expr;
    s += ");\n}\n#define IGX_F igx_f(x, y, z, pi)\n";
    s += RTC_LV12_BODY;
    return s;
}

static void fill_co(const igx_patch *pt, bool parametric, IgxCo &co)
{
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
}

// [G0][G1][G2] weight field x f(expr) -> [G0][N1][N2] (d_t2), by the generated kernel; IGX_ERR_UNSUPPORTED where k_lv12 does not
// apply (2D, long lines, unequal degrees of the last two axes): the caller takes the two-array path
int lv12_expr_applicable(const igx_patch *pt, int parametric)
{
    int clen, nch;
    size_t lds12;
    if (pt->boxed || !lv12_shape(pt, &clen, &nch, &lds12)) { set_error("load vector of an expression: the fused contraction kernel does not serve this patch"); return IGX_ERR_UNSUPPORTED; }
    if (!parametric && pt->geo_kind == IGX_GEO_JACOBIAN) { set_error("a function of the physical coordinates needs a spline geometry"); return IGX_ERR_UNSUPPORTED; }
    RtcApi *api;
    return rtc_api(&api);
}

int launch_lv12_expr(hipStream_t st, igx_patch *pt, const char *expr, int parametric, const double *d_W, double *d_t2, int *hit)
{
    int clen, nch;
    size_t lds12;
    if (pt->boxed || !lv12_shape(pt, &clen, &nch, &lds12)) { set_error("load vector of an expression: the fused contraction kernel does not serve this patch"); return IGX_ERR_UNSUPPORTED; }
    if (!parametric && pt->geo_kind == IGX_GEO_JACOBIAN) { set_error("a function of the physical coordinates needs a spline geometry"); return IGX_ERR_UNSUPPORTED; }
    const PatchDev &pd = pt->dev;
    const AxisDev a1 = pd.ax[1], a2 = pd.ax[2];
    const int npass = std::max(2, (a2.N + 63) / 64);
    hipFunction_t fn;
    if (int rc = exprs_ok(1, &expr, "igx_load_vector_expr")) return rc;
    if (int rc = rtc_function(pt, lv12_source(a2.P, npass, parametric != 0, expr), "igx_lv12_expr", &fn, hit)) return rc;
    IgxCo co{};
    fill_co(pt, parametric != 0, co);
    int G0 = pd.G0_loc;
    IGX_HIP(hipMemsetAsync(d_t2, 0, (size_t)G0 * a1.N * a2.N * sizeof(double), st));
    const long long units = (long long)G0 * nch;
    void *args[] = {(void *)&co, (void *)&d_W, (void *)&d_t2, (void *)&a1, (void *)&a2, (void *)&G0, (void *)&clen, (void *)&nch};
    if (hipModuleLaunchKernel(fn, (unsigned)((units + 3) / 4), 1, 1, 256, 1, 1, (unsigned)lds12, st, args, nullptr) != hipSuccess) {
        (void)hipGetLastError(); set_error("launch of the compiled load-vector kernel failed"); return IGX_ERR_HIP;
    }
    return IGX_OK;
}

int rtc_compile_lv12(int P, int npass, int parametric, const char *expr, const char *arch, char *path_out, int path_len, int *hit)
{
    if (!expr || !arch || P < 2 || P > 6 || npass < 2 || npass > 4) { set_error("igx_rtc_compile_load_vector: bad argument"); return IGX_ERR_ARG; }
    std::vector<char> code;
    std::string path;
    if (int rc = exprs_ok(1, &expr, "igx_rtc_compile_load_vector")) return rc;
    if (int rc = rtc_code_object(lv12_source(P, npass, parametric != 0, expr), arch, code, path, hit)) return rc;
    if (path_out && path_len > 0) { strncpy(path_out, path.c_str(), (size_t)path_len - 1); path_out[path_len - 1] = 0; }
    return IGX_OK;
}

int rtc_compile_form(int n_expr, const char *const *expr, const char *arch, char *path_out, int path_len, int *hit)
{
    if (!expr || !arch || n_expr < 1) { set_error("igx_rtc_compile_form: bad argument"); return IGX_ERR_ARG; }
    std::vector<char> code;
    std::string path;
    if (int rc = exprs_ok(n_expr, expr, "igx_rtc_compile_form")) return rc;
    if (int rc = rtc_code_object(form_source(n_expr, expr), arch, code, path, hit)) return rc;
    if (path_out && path_len > 0) { strncpy(path_out, path.c_str(), (size_t)path_len - 1); path_out[path_len - 1] = 0; }
    return IGX_OK;
}

} // namespace igx
