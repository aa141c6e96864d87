/* TEST INFRASTRUCTURE.  Glue for oracle/_ref/libref_host.so: no-op GLFW symbols
 * for the reference's camera.c and exported wrappers around the static inline
 * helpers of the reference's math_utilities.h (included from the reference
 * checkout at build time, never copied). */
#include <GLFW/glfw3.h>
#include "math_utilities.h"

int glfwGetKey(GLFWwindow* window, int key) { (void) window; (void) key; return GLFW_RELEASE; }
int glfwGetMouseButton(GLFWwindow* window, int button) { (void) window; (void) button; return GLFW_RELEASE; }
void glfwGetCursorPos(GLFWwindow* window, double* x, double* y) { (void) window; *x = 0.0; *y = 0.0; }
double glfwGetTime(void) { return 0.0; }

void ref_matrix_inverse(float inverse[4][4], const float matrix[4][4]) { matrix_inverse(inverse, matrix); }
uint32_t ref_wang_random_number(uint32_t seed) { return wang_random_number(seed); }
float ref_half_to_float(uint16_t half) { return half_to_float(half); }
