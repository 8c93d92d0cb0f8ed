/* A non-Python host of the hot path: loads a program exported by pgtformer_amd/export.py, restores the windows of a raw uint8
 * frame file and writes the restored uint8 frames - libpgt_hip.so + the HIP runtime only (tests/test_gpu_model.py::
 * test_exported_program_replays_bit_equal_from_python_and_from_c compares the result with the Python host's, bit for bit).
 *   gcc -D__HIP_PLATFORM_AMD__ tests/c/program_smoke.c -I include -I /opt/rocm/include -L pgtformer_amd/lib -lpgt_hip \
 *       -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/pgtformer_amd/lib -Wl,-rpath,/opt/rocm/lib -o program_smoke
 *   ./program_smoke model.prog frames_in.u8 frames_out.u8 [repeats]                                                        */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "pgt_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s model.prog in.u8 out.u8 [repeats]\n", argv[0]); return 1; }
    const int repeats = argc > 4 ? atoi(argv[4]) : 1;
    pgt_program* prog = NULL;
    if (pgt_program_load(argv[1], &prog) != 0) { fprintf(stderr, "load: %s\n", pgt_last_error()); return 3; }
    size_t nin = 0, nout = 0;
    pgt_program_io_bytes(prog, &nin, &nout);
    const size_t nws = pgt_program_workspace_bytes(prog);
    printf("%s\nprogram: %s\ninput %zu bytes, output %zu bytes, workspace %zu bytes\n", pgt_version(), pgt_program_info(prog), nin, nout, nws);
    unsigned char* hin = (unsigned char*)malloc(nin);
    unsigned char* hout = (unsigned char*)malloc(nout);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(hin, 1, nin, f) != nin) { fprintf(stderr, "cannot read %zu bytes from %s\n", nin, argv[2]); return 4; }
    fclose(f);
    void *din = NULL, *dout = NULL, *ws = NULL;
    hipStream_t st;
    CK(hipMalloc(&din, nin));
    CK(hipMalloc(&dout, nout));
    CK(hipMalloc(&ws, nws ? nws : 16));
    CK(hipStreamCreate(&st));
    CK(hipMemcpyAsync(din, hin, nin, hipMemcpyHostToDevice, st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < repeats; ++r) {
        if (r == repeats - 1) CK(hipEventRecord(e0, st));
        if (pgt_program_run(prog, din, dout, ws, nws, st) != 0) { fprintf(stderr, "run: %s\n", pgt_last_error()); return 5; }
    }
    CK(hipEventRecord(e1, st));
    CK(hipMemcpyAsync(hout, dout, nout, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("forward: %.3f ms (eager launches from C, run %d of %d)\n", ms, repeats, repeats);
    f = fopen(argv[3], "wb");
    if (!f || fwrite(hout, 1, nout, f) != nout) { fprintf(stderr, "cannot write %s\n", argv[3]); return 6; }
    fclose(f);
    pgt_program_destroy(prog);
    hipFree(din); hipFree(dout); hipFree(ws);
    free(hin); free(hout);
    return 0;
}
