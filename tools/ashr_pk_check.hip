// Does v_ashr_pk_u8_i32 (gfx950) compute what this compiler's pattern assumes?  The C expression below is selected to that instruction (check with -S);
// the host evaluates the same expression.  hipcc --offload-arch=gfx950 -O3 tools/ashr_pk_check.hip -o tools/bin/ashr_pk_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
__host__ __device__ inline unsigned int pack2(int a, int b, int sh) {
    const int ta = a >> sh, tb = b >> sh;
    const unsigned int ua = (unsigned int)(ta < 0 ? 0 : ta > 255 ? 255 : ta), ub = (unsigned int)(tb < 0 ? 0 : tb > 255 ? 255 : tb);
    return ua | (ub << 8);
}
__global__ void k(const int* a, const int* b, unsigned int* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pack2(a[i], b[i], 5);
}
int main() {
    const int n = 1 << 16;
    std::vector<int> a(n), b(n);
    std::mt19937 g(3);
    for (int i = 0; i < n; ++i) { a[i] = (int)(g() % 20000) - 6000; b[i] = (int)(g() % 20000) - 6000; }
    int *da, *db; unsigned int* dout;
    (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, n * 4); (void)hipMalloc(&dout, n * 4);
    (void)hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
    std::vector<unsigned int> out(n);
    (void)hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0, swapped = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned int want = pack2(a[i], b[i], 5), sw = pack2(b[i], a[i], 5);
        if (out[i] != want) { if (bad < 6) printf("a=%d b=%d: device 0x%04x, expected 0x%04x (operands exchanged: 0x%04x)\n", a[i], b[i], out[i], want, sw); ++bad; swapped += out[i] == sw; }
    }
    printf("%d of %d differ from the C expression; %d of those equal the expression with the operands exchanged\n", bad, n, swapped);
    return 0;
}
