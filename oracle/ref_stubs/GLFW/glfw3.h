/* TEST INFRASTRUCTURE.  Minimal stand-in for <GLFW/glfw3.h> so that the
 * reference's src/camera.c compiles unmodified for oracle/_ref (only its matrix
 * helpers are called; the input handler control_camera() links against the
 * no-op stubs in ref_host_shim.c). */
#ifndef ORACLE_GLFW_STUB_H
#define ORACLE_GLFW_STUB_H
typedef struct GLFWwindow GLFWwindow;
#define GLFW_RELEASE 0
#define GLFW_PRESS 1
#define GLFW_KEY_A 65
#define GLFW_KEY_D 68
#define GLFW_KEY_E 69
#define GLFW_KEY_Q 81
#define GLFW_KEY_S 83
#define GLFW_KEY_W 87
#define GLFW_KEY_LEFT_SHIFT 340
#define GLFW_KEY_LEFT_CONTROL 341
#define GLFW_MOUSE_BUTTON_2 1
int glfwGetKey(GLFWwindow* window, int key);
int glfwGetMouseButton(GLFWwindow* window, int button);
void glfwGetCursorPos(GLFWwindow* window, double* x, double* y);
double glfwGetTime(void);
#endif
