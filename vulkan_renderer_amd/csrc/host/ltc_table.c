/* LTC table loader.  File format and quantisation follow reference
 * src/ltc_table.c:23-194; the Vulkan texture arrays are replaced by two linear
 * device buffers. */
#include "vkr_internal.h"

static uint16_t quantize_unorm16(float value) {
	if (value < 0.0f) value = 0.0f;
	if (value > 1.0f) value = 1.0f;
	return (uint16_t) (value * 65535.0f + 0.5f);
}

int load_ltc_table(ltc_table_t* table, const device_t* device, const char* directory, uint32_t fresnel_count) {
	memset(table, 0, sizeof(*table));
	table->fresnel_count = fresnel_count;
	size_t texels_per_slice = 0;
	for (uint32_t slice = 0; slice != fresnel_count; ++slice) {
		char index_string[16];
		sprintf(index_string, "%u", slice);
		const char* pieces[] = {directory, "/fit", index_string, ".dat"};
		char* path = vkr_concatenate(VKR_COUNT_OF(pieces), pieces);
		FILE* file = fopen(path, "rb");
		if (!file) {
			printf("Failed to open the linearly transformed cosine table at %s.\n", path);
			free(path);
			destroy_ltc_table(table, device);
			return 1;
		}
		free(path);
		uint64_t resolution = 0;
		if (fread(&resolution, sizeof(resolution), 1, file) != 1 || resolution == 0 || resolution > 4096) {
			printf("The linearly transformed cosine table %u in directory %s has an invalid header.\n", slice, directory);
			fclose(file);
			destroy_ltc_table(table, device);
			return 1;
		}
		if (table->roughness_count == 0) {
			table->roughness_count = table->inclination_count = (uint32_t) resolution;
			texels_per_slice = (size_t) resolution * resolution;
			table->host_rgba = (uint16_t*) malloc(sizeof(uint16_t) * 4 * texels_per_slice * fresnel_count);
			table->host_rg = (uint16_t*) malloc(sizeof(uint16_t) * 2 * texels_per_slice * fresnel_count);
		}
		else if (resolution != table->roughness_count) {
			printf("The linearly transformed cosine tables in directory %s have inconsistent resolutions. One has resolution %llux%llu, another %ux%u.\n",
				directory, (unsigned long long) resolution, (unsigned long long) resolution, table->roughness_count, table->roughness_count);
			fclose(file);
			destroy_ltc_table(table, device);
			return 1;
		}
		uint16_t* rgba = table->host_rgba + 4 * texels_per_slice * slice;
		uint16_t* rg = table->host_rg + 2 * texels_per_slice * slice;
		for (size_t texel = 0; texel != texels_per_slice; ++texel) {
			/* four free entries of the cosine-to-shading matrix and the albedo */
			float fit[5];
			if (fread(fit, sizeof(float), 5, file) != 5) {
				printf("The linearly transformed cosine table %u in directory %s is truncated.\n", slice, directory);
				fclose(file);
				destroy_ltc_table(table, device);
				return 1;
			}
			/* adjugate (inverse up to a factor), entries as at ltc_table.c:86-90 */
			float adj[3][3] = {
				{fit[2], 0.0f, -fit[1] * fit[2]},
				{0.0f, fit[0] - fit[1] * fit[3], 0.0f},
				{-fit[2] * fit[3], 0.0f, fit[0] * fit[2]}
			};
			float largest = fabsf(adj[0][0]);
			for (uint32_t r = 0; r != 3; ++r)
				for (uint32_t c = 0; c != 3; ++c)
					if (largest < fabsf(adj[r][c])) largest = fabsf(adj[r][c]);
			for (uint32_t r = 0; r != 3; ++r)
				for (uint32_t c = 0; c != 3; ++c)
					adj[r][c] /= largest;
			rgba[4 * texel + 0] = quantize_unorm16(adj[0][0]);
			rgba[4 * texel + 1] = quantize_unorm16(adj[0][2] * -1.0f);
			rgba[4 * texel + 2] = quantize_unorm16(adj[1][1]);
			rgba[4 * texel + 3] = quantize_unorm16(adj[2][0]);
			rg[2 * texel + 0] = quantize_unorm16(adj[2][2]);
			rg[2 * texel + 1] = quantize_unorm16(fit[4]);
		}
		fclose(file);
	}
	if (device) {
		size_t total = texels_per_slice * fresnel_count;
		if (vkr_device_upload(&table->device_rgba, device, table->host_rgba, sizeof(uint16_t) * 4 * total, "LTC tables (RGBA16)")
			|| vkr_device_upload(&table->device_rg, device, table->host_rg, sizeof(uint16_t) * 2 * total, "LTC tables (RG16)"))
		{
			destroy_ltc_table(table, device);
			return 1;
		}
	}
	/* lookup constants, ltc_table.c:184-191 */
	table->constants.fresnel_index_factor = (float) (table->fresnel_count - 1);
	table->constants.fresnel_index_summand = 0.0f;
	table->constants.roughness_factor = (float) (table->roughness_count - 1) / (float) table->roughness_count;
	table->constants.roughness_summand = 0.5f / (float) table->roughness_count;
	table->constants.inclination_factor = (float) (table->inclination_count - 1) / (0.5f * VKR_PI_F * table->inclination_count);
	table->constants.inclination_summand = 0.5f / (float) table->inclination_count;
	return 0;
}

void destroy_ltc_table(ltc_table_t* table, const device_t* device) {
	free(table->host_rgba);
	free(table->host_rg);
	vkr_device_free(table->device_rgba, device);
	vkr_device_free(table->device_rg, device);
	memset(table, 0, sizeof(*table));
}
