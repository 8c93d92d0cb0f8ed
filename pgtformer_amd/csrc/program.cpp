// Whole-graph entry: a recorded forward (pgtformer_amd/export.py) replayed by the library itself - what SURVEY.md section 8b calls
// `pgt_forward_window` with an opaque model handle: the orchestration of the reference's PGTFormer.forward
// (archs/pgtformer_arch.py:598-714, driver semantics inference.py:12-19) as a TAPE of C-ABI calls.
//
// A program file holds (export.py write_program): the names of the tape functions, sizes, the calls (function id + arguments:
// integers, floats, descriptor structs, pointers as (region, offset)), and the bytes of the PERSISTENT regions (repacked weights,
// tables, index tensors, zeroed arrival counters).  pgt_program_load uploads the persistent bytes once (the only allocation);
// pgt_program_run resolves every pointer against {persistent block, caller's workspace, caller's input, caller's output} and makes
// the calls in order on the caller's stream: no allocation, no synchronisation, no host-side shape logic - a C host may capture it
// into a hipGraph.  Errors: negative errno-style codes, message via pgt_last_error().
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "pgt_internal.h"

namespace {

union TapeArg {
    int64_t i;
    float f;
    void* p;
};
struct AnyPtr {      // a tape pointer converts to whatever pointer type the parameter has
    void* p;
    template <class T> operator T*() const { return reinterpret_cast<T*>(p); }
};

#include "program_dispatch.inc"

enum { K_INT = 0, K_F32 = 1, K_NULL = 2, K_PTR = 3, K_DESC = 4, K_STREAM = 5 };
enum { R_PERSIST = 0, R_WORK = 1, R_IN = 2, R_OUT = 3 };

struct ArgRec {
    uint32_t kind, aux;
    uint64_t value;
};
struct Call {
    int fn;            // index into the library's dispatch table
    uint32_t first, n; // its ArgRecs
};
struct Program {
    uint64_t persist_bytes = 0, work_bytes = 0, in_bytes = 0, out_bytes = 0;
    std::vector<Call> calls;
    std::vector<ArgRec> args;
    std::vector<char> pool;        // descriptor structs
    char* persist = nullptr;       // device
    std::string meta;
};

bool rd(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n; }

}  // namespace

extern "C" int pgt_program_load(const char* path, pgt_program** out) {
    PGT_CHECK(path && out, "pgt_program_load: null argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    PGT_CHECK(f != nullptr, "pgt_program_load: cannot open %s", path);
    Program* pr = new Program();
    auto fail = [&](const char* what) {
        pgt_set_error("pgt_program_load: %s (%s)", what, path);
        fclose(f);
        if (pr->persist) (void)hipFree(pr->persist);
        delete pr;
        return -22;
    };
    char magic[8];
    uint32_t version = 0, nfn = 0;
    if (!rd(f, magic, 8) || memcmp(magic, "PGTPROG1", 8) != 0) return fail("not a program file");
    if (!rd(f, &version, 4) || !rd(f, &nfn, 4) || version != 1 || nfn > 4096) return fail("unsupported version");
    // the file's function table -> this library's dispatch ids (by NAME: a library with more functions still runs older tapes)
    std::vector<int> map(nfn, -1);
    for (uint32_t i = 0; i < nfn; ++i) {
        uint16_t len = 0;
        char name[256];
        if (!rd(f, &len, 2) || len >= sizeof(name) || !rd(f, name, len)) return fail("truncated function table");
        name[len] = 0;
        for (int k = 0; k < kTapeFunctions; ++k)
            if (strcmp(kTapeNames[k], name) == 0) map[i] = k;
    }
    uint32_t ncalls = 0, meta_len = 0;
    if (!rd(f, &pr->persist_bytes, 8) || !rd(f, &pr->work_bytes, 8) || !rd(f, &pr->in_bytes, 8) || !rd(f, &pr->out_bytes, 8) ||
        !rd(f, &ncalls, 4) || !rd(f, &meta_len, 4) || meta_len > (1u << 20))
        return fail("truncated header");
    pr->meta.resize(meta_len);
    if (meta_len && !rd(f, &pr->meta[0], meta_len)) return fail("truncated header");
    pr->calls.reserve(ncalls);
    for (uint32_t c = 0; c < ncalls; ++c) {
        uint16_t fid = 0, na = 0;
        if (!rd(f, &fid, 2) || !rd(f, &na, 2) || fid >= nfn || na > 32) return fail("bad call record");
        if (map[fid] < 0) return fail("the tape calls a function this library does not have");
        Call cl{map[fid], (uint32_t)pr->args.size(), na};
        // the record against the parameter list of the function it names: count and class of every argument (a descriptor
        // is host memory, a pointer device memory, an integer must not land in a pointer parameter)
        const char* sig = kTapeSig[cl.fn];
        if (strlen(sig) != na) return fail("argument count of another library version");
        for (uint16_t k = 0; k < na; ++k) {
            ArgRec r;
            if (!rd(f, &r.kind, 4) || !rd(f, &r.aux, 4) || !rd(f, &r.value, 8)) return fail("bad argument record");
            const char want = sig[k];
            const bool ok = r.kind == K_INT ? want == 'i' : r.kind == K_F32 ? want == 'f' : r.kind == K_DESC ? want == 'd' :
                            r.kind == K_STREAM ? want == 's' : (r.kind == K_NULL || r.kind == K_PTR) ? want == 'p' : false;
            if (!ok) return fail("argument of the wrong class for its parameter");
            pr->args.push_back(r);
        }
        pr->calls.push_back(cl);
    }
    uint64_t pool_bytes = 0;
    if (!rd(f, &pool_bytes, 8) || pool_bytes > (1ull << 30)) return fail("bad descriptor pool");
    pr->pool.resize(pool_bytes);
    if (pool_bytes && !rd(f, pr->pool.data(), pool_bytes)) return fail("truncated descriptor pool");
    // every argument record is checked once here: descriptor size (another library version), and the START of every pointer against its
    // region.  The EXTENT a call touches follows from the integer arguments and descriptors the writer recorded and is the called
    // function's business (its own argument checks): the loader guards against truncation and version skew, not against a hostile
    // file - pgtformer_amd/export.py only keeps a file whose replay reproduced the recorded frames bit for bit
    for (const ArgRec& r : pr->args) {
        if (r.kind == K_DESC && (r.aux != sizeof(pgt_conv_desc) || r.value > pool_bytes || pool_bytes - r.value < r.aux || r.value % 8 != 0))
            return fail("descriptor of another library version");
        if (r.kind == K_PTR) {
            const uint64_t lim = r.aux == R_PERSIST ? pr->persist_bytes : r.aux == R_WORK ? pr->work_bytes : r.aux == R_IN ? pr->in_bytes : r.aux == R_OUT ? pr->out_bytes : 0;
            if (r.value >= lim) return fail("pointer outside its region");
        }
    }
    if (pr->persist_bytes) {
        if (hipMalloc((void**)&pr->persist, pr->persist_bytes) != hipSuccess) return fail("cannot allocate the persistent block");
        std::vector<char> buf(64u << 20);
        for (uint64_t done = 0; done < pr->persist_bytes;) {
            const size_t n = (size_t)((pr->persist_bytes - done) < buf.size() ? (pr->persist_bytes - done) : buf.size());
            if (!rd(f, buf.data(), n)) return fail("truncated persistent data");
            if (hipMemcpy(pr->persist + done, buf.data(), n, hipMemcpyHostToDevice) != hipSuccess) return fail("upload failed");
            done += n;
        }
    }
    fclose(f);
    *out = reinterpret_cast<pgt_program*>(pr);
    return 0;
}

extern "C" void pgt_program_destroy(pgt_program* h) {
    Program* pr = reinterpret_cast<Program*>(h);
    if (!pr) return;
    if (pr->persist) (void)hipFree(pr->persist);
    delete pr;
}

extern "C" size_t pgt_program_workspace_bytes(const pgt_program* h) {
    return h ? (size_t)reinterpret_cast<const Program*>(h)->work_bytes : 0;
}

extern "C" int pgt_program_io_bytes(const pgt_program* h, size_t* in_bytes, size_t* out_bytes) {
    PGT_CHECK(h && in_bytes && out_bytes, "pgt_program_io_bytes: null argument");
    *in_bytes = (size_t)reinterpret_cast<const Program*>(h)->in_bytes;
    *out_bytes = (size_t)reinterpret_cast<const Program*>(h)->out_bytes;
    return 0;
}

extern "C" const char* pgt_program_info(const pgt_program* h) {
    return h ? reinterpret_cast<const Program*>(h)->meta.c_str() : "";
}

extern "C" int pgt_program_run(const pgt_program* h, const void* input, void* output, void* workspace, size_t workspace_bytes,
                               pgt_stream_t stream) {
    const Program* pr = reinterpret_cast<const Program*>(h);
    PGT_CHECK(pr && input && output, "pgt_program_run: null argument");
    PGT_CHECK(workspace_bytes >= pr->work_bytes && (workspace || pr->work_bytes == 0) && (((uintptr_t)workspace | (uintptr_t)input | (uintptr_t)output) & 15) == 0,
              "pgt_program_run: workspace of %zu bytes (need %llu), buffers 16-byte aligned", workspace_bytes, (unsigned long long)pr->work_bytes);
    char* const base[4] = {pr->persist, (char*)workspace, (char*)const_cast<void*>(input), (char*)output};
    TapeArg a[32];
    for (size_t c = 0; c < pr->calls.size(); ++c) {
        const Call& cl = pr->calls[c];
        for (uint32_t k = 0; k < cl.n; ++k) {
            const ArgRec& r = pr->args[cl.first + k];
            switch (r.kind) {
                case K_INT: a[k].i = (int64_t)r.value; break;
                case K_F32: { const uint32_t b = (uint32_t)r.value; memcpy(&a[k].f, &b, 4); break; }
                case K_NULL: a[k].p = nullptr; break;
                case K_PTR: a[k].p = base[r.aux] + r.value; break;
                case K_DESC: a[k].p = const_cast<char*>(pr->pool.data()) + r.value; break;
                default: a[k].p = (void*)stream; break;
            }
        }
        const int rc = call_tape_function(cl.fn, a);
        if (rc != 0) return rc;      // (pgt_last_error holds the failing function's message)
    }
    return 0;
}
