// C entry points around the reference's own create_camera /
// camera_set_distance (declared in the reference's src/camera.hpp, which this
// file includes from the reference tree at build time; no reference source is
// copied here). Linked together with the reference's src/camera.cpp into
// oracle/_ref/libref_camera.so by oracle/Makefile.
#include "camera.hpp"

#include <cstring>

static_assert(sizeof(Camera) == 84, "Camera is 21 floats");

extern "C" void ref_create_camera(const float position[3],
                                  const float target[3],
                                  float sensor_distance,
                                  float sensor_half_width,
                                  float sensor_half_height, float out[21])
{
    const Camera camera = create_camera(
        vec3 {position[0], position[1], position[2]},
        vec3 {target[0], target[1], target[2]}, sensor_distance,
        sensor_half_width, sensor_half_height);
    std::memcpy(out, &camera, sizeof camera);
}

extern "C" void ref_camera_set_distance(float camera_io[21], float distance)
{
    Camera camera;
    std::memcpy(&camera, camera_io, sizeof camera);
    camera_set_distance(camera, distance);
    std::memcpy(camera_io, &camera, sizeof camera);
}
