/* Tabulated noise: reference src/noise_table.h:21-89. */
#ifndef VKR_NOISE_TABLE_H
#define VKR_NOISE_TABLE_H
#include "vkr_device.h"

/*! Same values as reference noise_table.h:21-55 */
typedef enum noise_type_e {
	noise_type_white = 0,
	noise_type_blue,
	noise_type_ahmed,
	noise_type_count,
	noise_type_sobol,
	noise_type_owen,
	noise_type_burley_owen,
	noise_type_blue_noise_dithered,
	noise_type_full_count,
} noise_type_t;

typedef struct noise_table_s {
	/*! RGBA16_UNORM, layer-major: depth * height * width * 4 uint16_t */
	VkExtent3D resolution;
	uint16_t* host_data;
	void* device_data;
	/*! Seed for the per-frame randomisation (reference noise_table.h:66) */
	uint32_t random_seed;
} noise_table_t;

/*! reference noise_table.h:71 / noise_table.c:23-43 */
VKR_API VkExtent3D get_default_noise_resolution(noise_type_t noise_type);
/*! reference noise_table.h:81 / noise_table.c:46-153.  White noise is generated;
	every other type is read from data/noise/<type>_..._<W>x<H>_<D>.blob relative to
	the working directory, exactly like the reference. */
VKR_API int load_noise_table(noise_table_t* noise, const device_t* device, VkExtent3D resolution, noise_type_t noise_type);
/*! reference noise_table.h:84 */
VKR_API void destroy_noise_table(noise_table_t* noise, const device_t* device);
/*! reference noise_table.h:89 / noise_table.c:161-168 */
VKR_API void set_noise_constants(uint32_t resolution_mask[2], uint32_t* texture_index_mask, uint32_t random_numbers[4], noise_table_t* noise, VkBool32 animate_noise);

#endif
