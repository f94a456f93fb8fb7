/* TEST INFRASTRUCTURE - not product code.
 *
 * NDRange driver for the host-compiled VERBATIM reference kernels (oracle/_ref).
 * Plays the role of OpenCLRaytracer::Render()'s enqueue_1d_range_kernel
 * (/root/reference/OpenCLRaytracer.cpp:89-91): calls the kernel entry once per
 * work-item with get_global_id(0) = i, OpenMP-parallel over work-items.
 *
 * One shared object per (kernel file, contraction flavour) because the three .cl
 * files define clashing globals (raycast / shade / MAX_FLOAT). Select the kernel with
 * -DREF_KERNEL=0 (hittest_kernel.cl:54) | 1 (shade_kernel.cl:180) |
 *  2 (shade_and_reflect_kernel.cl:244).
 */
#include <stdint.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread uint64_t g_global_id;

/* size_t get_global_id(uint) as mangled by clang's OpenCL front-end */
uint64_t ref_get_global_id(uint32_t dim) __asm__("_Z13get_global_idj");
uint64_t ref_get_global_id(uint32_t dim) { (void)dim; return g_global_id; }

#if REF_KERNEL == 0
extern void hittest(uint64_t count, const void* objs, const void* rays, float* hits);
#elif REF_KERNEL == 1
extern void shade(uint32_t max_bounces, uint32_t n_objs, const void* objs, uint32_t n_lights,
                  const void* lights, const void* rays, void* pixels);
#else
extern void shade_and_reflect(uint32_t max_bounces, uint32_t n_objs, const void* objs, uint32_t n_lights,
                              const void* lights, const void* rays, void* pixels);
#endif

int ref_kernel_id(void) { return REF_KERNEL; }

/* Run work-items [first, first+count). `out` is indexed by global id exactly like the
 * reference's pixelData / hits buffers (16-B stride for pixels, 4-B for hits); elements of
 * work-items that miss are left untouched, as in the reference. Returns threads used. */
int ref_run(uint32_t max_bounces, uint32_t n_objs, const void* objs, uint32_t n_lights, const void* lights,
            const void* rays, void* out, uint64_t first, uint64_t count, int threads)
{
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#pragma omp parallel for schedule(dynamic, 64) num_threads(threads)
#endif
    for (int64_t i = (int64_t)first; i < (int64_t)(first + count); ++i) {
        g_global_id = (uint64_t)i;
#if REF_KERNEL == 0
        (void)max_bounces; (void)n_lights; (void)lights;
        hittest((uint64_t)n_objs, objs, rays, (float*)out);
#elif REF_KERNEL == 1
        shade(max_bounces, n_objs, objs, n_lights, lights, rays, out);
#else
        shade_and_reflect(max_bounces, n_objs, objs, n_lights, lights, rays, out);
#endif
    }
    return used;
}
