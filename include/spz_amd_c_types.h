/* spz_amd_c_types.h — the plain-C view of a GaussianCloud used to hand clouds to C, Swift or C# hosts.
 * Same type names, field names, field order and ownership rule as the reference's bridge header
 * (/root/reference/src/cc/splat-c-types.h:11-27, filled by GaussianCloud::data(), splat-types.h:117-130):
 * every buffer is allocated with new float[] by the library and freed by the caller. */
#ifndef SPZ_AMD_C_TYPES_H_
#define SPZ_AMD_C_TYPES_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

typedef struct {
  size_t count; /* number of floats */
  float *data;  /* NULL when count == 0 */
} SpzFloatBuffer;

typedef struct {
  int32_t numPoints;
  int32_t shDegree;
  bool antialiased;
  SpzFloatBuffer positions;
  SpzFloatBuffer scales;
  SpzFloatBuffer rotations;
  SpzFloatBuffer alphas;
  SpzFloatBuffer colors;
  SpzFloatBuffer sh;
} GaussianCloudData;

#endif /* SPZ_AMD_C_TYPES_H_ */
