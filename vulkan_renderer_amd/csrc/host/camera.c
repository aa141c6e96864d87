/* Camera matrices, pure C.  Same arithmetic as reference src/camera.c:24-83. */
#include "vkr_internal.h"

void get_world_to_view_space(float world_to_view_space[4][4], const first_person_camera_t* camera) {
	float cx = cosf(camera->rotation_x), sx = sinf(camera->rotation_x);
	float cz = cosf(camera->rotation_z), sz = sinf(camera->rotation_z);
	float about_x[3][3] = {{1.0f, 0.0f, 0.0f}, {0.0f, cx, sx}, {0.0f, -sx, cx}};
	float about_z[3][3] = {{cz, sz, 0.0f}, {-sz, cz, 0.0f}, {0.0f, 0.0f, 1.0f}};
	/* view-to-world rotation = about_z * about_x */
	float rot[3][3];
	for (uint32_t i = 0; i != 3; ++i)
		for (uint32_t j = 0; j != 3; ++j) {
			float sum = 0.0f;
			for (uint32_t l = 0; l != 3; ++l) sum += about_z[i][l] * about_x[l][j];
			rot[i][j] = sum;
		}
	/* where the world origin ends up in view space */
	float origin[3];
	for (uint32_t i = 0; i != 3; ++i) {
		float sum = 0.0f;
		for (uint32_t j = 0; j != 3; ++j) sum -= rot[j][i] * camera->position_world_space[j];
		origin[i] = sum;
	}
	for (uint32_t i = 0; i != 3; ++i) {
		for (uint32_t j = 0; j != 3; ++j) world_to_view_space[i][j] = rot[j][i];
		world_to_view_space[i][3] = origin[i];
	}
	world_to_view_space[3][0] = world_to_view_space[3][1] = world_to_view_space[3][2] = 0.0f;
	world_to_view_space[3][3] = 1.0f;
}

void get_view_to_projection_space(float view_to_projection_space[4][4], const first_person_camera_t* camera, float aspect_ratio) {
	float near = camera->near, far = camera->far;
	float top = tanf(0.5f * camera->vertical_fov);
	float right = aspect_ratio * top;
	memset(view_to_projection_space, 0, sizeof(float) * 16);
	view_to_projection_space[0][0] = -1.0f / right;
	view_to_projection_space[1][1] = 1.0f / top;
	view_to_projection_space[2][2] = -(far + near) / (far - near);
	view_to_projection_space[2][3] = -2.0f * far * near / (far - near);
	view_to_projection_space[3][2] = -1.0f;
}

void get_world_to_projection_space(float world_to_projection_space[4][4], const first_person_camera_t* camera, float aspect_ratio) {
	float w2v[4][4], v2p[4][4];
	get_world_to_view_space(w2v, camera);
	get_view_to_projection_space(v2p, camera, aspect_ratio);
	for (uint32_t i = 0; i != 4; ++i)
		for (uint32_t j = 0; j != 4; ++j) {
			float sum = 0.0f;
			for (uint32_t l = 0; l != 4; ++l) sum += v2p[i][l] * w2v[l][j];
			world_to_projection_space[i][j] = sum;
		}
}
