/* TEST INFRASTRUCTURE.  Opaque stand-ins for the Vulkan types that the reference's
 * headers (main.h and what it includes) mention, so that host-only translation units
 * of the reference (experiment_list.c) compile unmodified for oracle/_ref.  Nothing
 * here is functional: handles are pointers, structs are opaque blobs, and only the
 * few plain structs whose members the reference's host code reads are spelled out. */
#ifndef ORACLE_VULKAN_TYPES_STUB_H
#define ORACLE_VULKAN_TYPES_STUB_H
#include <stdint.h>
typedef uint32_t VkBool32;
typedef uint64_t VkDeviceSize;
typedef uint32_t VkFlags;
#define VK_TRUE 1u
#define VK_FALSE 0u
#define VK_WHOLE_SIZE (~0ull)
#define ORACLE_VK_HANDLE(name) typedef struct name##_T* name;
ORACLE_VK_HANDLE(VkInstance) ORACLE_VK_HANDLE(VkPhysicalDevice) ORACLE_VK_HANDLE(VkDevice) ORACLE_VK_HANDLE(VkQueue)
ORACLE_VK_HANDLE(VkSemaphore) ORACLE_VK_HANDLE(VkCommandBuffer) ORACLE_VK_HANDLE(VkFence) ORACLE_VK_HANDLE(VkDeviceMemory)
ORACLE_VK_HANDLE(VkBuffer) ORACLE_VK_HANDLE(VkImage) ORACLE_VK_HANDLE(VkBufferView) ORACLE_VK_HANDLE(VkImageView)
ORACLE_VK_HANDLE(VkShaderModule) ORACLE_VK_HANDLE(VkPipelineLayout) ORACLE_VK_HANDLE(VkRenderPass) ORACLE_VK_HANDLE(VkPipeline)
ORACLE_VK_HANDLE(VkDescriptorSetLayout) ORACLE_VK_HANDLE(VkSampler) ORACLE_VK_HANDLE(VkDescriptorPool) ORACLE_VK_HANDLE(VkDescriptorSet)
ORACLE_VK_HANDLE(VkFramebuffer) ORACLE_VK_HANDLE(VkCommandPool) ORACLE_VK_HANDLE(VkSurfaceKHR) ORACLE_VK_HANDLE(VkSwapchainKHR)
ORACLE_VK_HANDLE(VkAccelerationStructureKHR)
typedef int VkImageLayout, VkFormat, VkPresentModeKHR, VkShaderStageFlagBits, VkMemoryPropertyFlagBits, VkMemoryHeapFlagBits, VkStructureType;
typedef VkFlags VkMemoryPropertyFlags, VkShaderStageFlags, VkBufferUsageFlags, VkImageUsageFlags;
enum { VK_IMAGE_LAYOUT_UNDEFINED = 0, VK_IMAGE_LAYOUT_TRANSFER_DST_OPTIMAL = 7,
	VK_IMAGE_USAGE_TRANSFER_SRC_BIT = 1, VK_IMAGE_USAGE_TRANSFER_DST_BIT = 2, VK_IMAGE_USAGE_SAMPLED_BIT = 4,
	VK_BUFFER_USAGE_TRANSFER_SRC_BIT = 1, VK_BUFFER_USAGE_TRANSFER_DST_BIT = 2, VK_MEMORY_PROPERTY_DEVICE_LOCAL_BIT = 1,
	VK_STRUCTURE_TYPE_WRITE_DESCRIPTOR_SET = 35, VK_STRUCTURE_TYPE_IMAGE_VIEW_CREATE_INFO = 15 };
typedef struct { uint32_t width, height; } VkExtent2D;
typedef struct { uint32_t width, height, depth; } VkExtent3D;
#define ORACLE_VK_BLOB(name) typedef struct { uint64_t opaque[128]; } name;
typedef struct { struct { VkDeviceSize nonCoherentAtomSize; } limits; uint64_t opaque[128]; } VkPhysicalDeviceProperties;
ORACLE_VK_BLOB(VkPhysicalDeviceMemoryProperties)
ORACLE_VK_BLOB(VkPhysicalDeviceAccelerationStructurePropertiesKHR) ORACLE_VK_BLOB(VkQueueFamilyProperties)
ORACLE_VK_BLOB(VkImageViewCreateInfo) ORACLE_VK_BLOB(VkImageCreateInfo) ORACLE_VK_BLOB(VkImageCopy) ORACLE_VK_BLOB(VkBufferImageCopy)
ORACLE_VK_BLOB(VkBufferCreateInfo) ORACLE_VK_BLOB(VkBufferCopy) ORACLE_VK_BLOB(VkDescriptorSetLayoutBinding)
ORACLE_VK_BLOB(VkWriteDescriptorSet) ORACLE_VK_BLOB(VkSurfaceFormatKHR) ORACLE_VK_BLOB(VkMemoryRequirements)
ORACLE_VK_BLOB(VkMappedMemoryRange) ORACLE_VK_BLOB(VkDescriptorImageInfo)
#endif
