/* Linearly transformed cosine tables: reference src/ltc_table.h:23-72.  The two
 * Vulkan texture arrays become two linear device buffers (RGBA16 and RG16
 * UNORM, layer-major); filtering is done in the kernel. */
#ifndef VKR_LTC_TABLE_H
#define VKR_LTC_TABLE_H
#include "vkr_device.h"

/*! Layout of reference ltc_table.h:23-35; copied into the constant buffer */
typedef struct ltc_constants_s {
	float fresnel_index_factor, fresnel_index_summand;
	float roughness_factor, roughness_summand;
	float inclination_factor, inclination_summand;
	float padding[2];
} ltc_constants_t;

typedef struct ltc_table_s {
	uint32_t roughness_count, inclination_count, fresnel_count;
	/*! Host copies: fresnel_count * inclination_count * roughness_count texels of
		4 (resp. 2) uint16_t.  Channels as in reference ltc_table.c:103-113. */
	uint16_t* host_rgba;
	uint16_t* host_rg;
	/*! Device copies (NULL when loaded without a device) */
	void* device_rgba;
	void* device_rg;
	ltc_constants_t constants;
} ltc_table_t;

/*! reference ltc_table.h:69 / ltc_table.c:23-194: reads <directory>/fit<i>.dat for
	i < fresnel_count, quantises and uploads.  Returns 0 on success; on failure
	prints the reason, cleans up and returns 1. */
VKR_API int load_ltc_table(ltc_table_t* table, const device_t* device, const char* directory, uint32_t fresnel_count);
/*! reference ltc_table.h:72 */
VKR_API void destroy_ltc_table(ltc_table_t* table, const device_t* device);

#endif
