/* First-person camera: layout of reference src/camera.h:27-47 (it is written raw
 * into quicksaves) and the three matrix helpers of src/camera.c:24-83.  The GLFW
 * input handler control_camera() has no counterpart. */
#ifndef VKR_CAMERA_H
#define VKR_CAMERA_H
#include "vkr_device.h"

typedef struct first_person_camera_s {
	float position_world_space[3];
	float rotation_z;
	float rotation_x;
	float vertical_fov;
	float near, far;
	float speed;
	int rotate_camera;
	float rotation_x_0, rotation_z_0;
} first_person_camera_t;

/*! reference camera.h:53 */
VKR_API void get_world_to_view_space(float world_to_view_space[4][4], const first_person_camera_t* camera);
/*! reference camera.h:57 */
VKR_API void get_view_to_projection_space(float view_to_projection_space[4][4], const first_person_camera_t* camera, float aspect_ratio);
/*! reference camera.h:61 */
VKR_API void get_world_to_projection_space(float world_to_projection_space[4][4], const first_person_camera_t* camera, float aspect_ratio);

#endif
