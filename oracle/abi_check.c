/* abi_check.c -- TEST INFRASTRUCTURE.  Compiled (never linked or run) by `make -C oracle ref` in the build
 * container: the reference's own driver headers and include/yolo2_hip.h in ONE translation unit.  C rejects
 * incompatible redeclarations, so this file compiling proves that every entry point libyolo2_hip.so exports
 * under a reference name has the reference's exact prototype (linux_app/include/yolo2_accel_linux.h:19-134,
 * dma_buffer_manager.h:24-139) and that the status codes agree (yolo2_config.h:146-151). */
#include "yolo2_config.h"
#include "yolo2_accel_linux.h"
#include "dma_buffer_manager.h"
#include "yolo2_hip.h"

_Static_assert(YOLO2_SUCCESS == 0 && YOLO2_ERROR == -1 && YOLO2_TIMEOUT == -2 && YOLO2_INIT_ERROR == -3 &&
               YOLO2_MMAP_ERROR == -4 && YOLO2_DMA_ERROR == -5, "status codes");
_Static_assert(Tm == 32 && Tn == 4, "tile constants");

/* every reference driver symbol, taken by address with the reference's own prototype */
int (*const p_init)(void) = yolo2_accel_init;
void (*const p_cleanup)(void) = yolo2_accel_cleanup;
void (*const p_setq)(int32_t, int32_t, int32_t, int32_t) = yolo2_set_q_values;
int (*const p_busy)(void) = yolo2_is_busy;
int (*const p_done)(void) = yolo2_is_done;
int (*const p_wait)(uint32_t) = yolo2_wait_for_completion;
uint32_t (*const p_status)(void) = yolo2_get_status;
uint32_t (*const p_rd)(uint32_t) = yolo2_read_reg;
void (*const p_wr)(uint32_t, uint32_t) = yolo2_write_reg;
int (*const p_dinit)(void) = dma_buffer_init;
void (*const p_dclean)(void) = dma_buffer_cleanup;
int (*const p_dalloc)(size_t, dma_buffer_t *) = dma_buffer_alloc;
void (*const p_dfree)(dma_buffer_t *) = dma_buffer_free;
void (*const p_dsd)(dma_buffer_t *, size_t, size_t) = dma_buffer_sync_for_device;
void (*const p_dsc)(dma_buffer_t *, size_t, size_t) = dma_buffer_sync_for_cpu;
uint64_t (*const p_dphys)(dma_buffer_t *, size_t) = dma_buffer_get_phys;
int (*const p_mddr)(size_t, size_t, memory_buffer_t *) = memory_allocate_ddr;
void (*const p_mfree)(memory_buffer_t *) = memory_free_ddr;
int (*const p_mw)(size_t, memory_buffer_t *) = memory_allocate_weights;
int (*const p_mb)(size_t, memory_buffer_t *) = memory_allocate_bias;
int (*const p_mi)(memory_buffer_t *) = memory_allocate_inference_buffer;
uint64_t (*const p_mphys)(void *) = memory_get_phys_addr;
void (*const p_mfl)(void *, size_t) = memory_flush_cache;
void (*const p_minv)(void *, size_t) = memory_invalidate_cache;
