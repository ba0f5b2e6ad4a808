/*
 * ref_image_as_buffer.cl -- TEST INFRASTRUCTURE ONLY.  Force-included (-include) in front of the reference's
 * src/GPU_kernels.cl for ONE of the two gfx950 builds of oracle/build_ref.sh.
 *
 * Why it exists: the MI355X has no image hardware.  Its OpenCL device reports CL_DEVICE_IMAGE_SUPPORT = 0 and
 * clCreateImage2D fails with CL_INVALID_OPERATION (seen on the GPU box, gpurun_out/r02a_*), so the reference's two
 * kernels that sample an image2d_t (luma_search_2step, prepare_predictors_and_residual; src/GPU_kernels.cl:562,
 * 1068, 1285) cannot run on this GPU as they stand.  For those two kernels -- and only in the build that serves
 * them -- the image argument becomes a __global uchar buffer and read_imageui becomes the texel fetch below:
 * unnormalised integer coordinates, CLK_ADDRESS_CLAMP_TO_EDGE, CLK_FILTER_NEAREST, CL_R / CL_UNSIGNED_INT8
 * (the sampler and format of src/GPU_kernels.cl:562 and src/init.h:559-578; OpenCL 1.2 section 8.2: coordinates
 * clamped to [0, w-1] x [0, h-1], result (texel, 0, 0, 1)).  Every other built-in the kernels call
 * (convert_*_sat, abs, select, mad24, vload/vstore, ...) is the vendor's device library in both builds, and the
 * eleven image-free kernels of GPU_kernels.cl plus everything of CPU_kernels.cl run from builds without this file.
 *
 * The host (oracle/ref_cl_driver.c) puts {width, height} in the 16 bytes in front of the pixel data.
 */
typedef __global const uchar *ref_image_buffer_t;
#define image2d_t ref_image_buffer_t
#define __read_only
#define read_only

uint4 __attribute__((overloadable)) ref_read_imageui_buffer(ref_image_buffer_t img, int2 c) {
    const int w = ((__global const int *)img)[-4], h = ((__global const int *)img)[-3];
    const int x = c.x < 0 ? 0 : (c.x > w - 1 ? w - 1 : c.x);
    const int y = c.y < 0 ? 0 : (c.y > h - 1 ? h - 1 : c.y);
    return (uint4)((uint)img[y * w + x], 0u, 0u, 1u);
}
#define read_imageui(img, smp, c) ref_read_imageui_buffer((img), (c))
