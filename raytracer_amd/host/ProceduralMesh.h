// Deterministic in-memory meshes standing in for the reference's .glb assets
// (res/models.7z is not available).  No reference counterpart; every generator
// produces the MeshData a glTF load would: light triangles first, per-vertex
// normals/uvs, model-local material and texture indices.
#pragma once
#include "MeshData.h"

namespace procedural {

// Cornell box [pos, pos+boxLen]^3 as triangles: ceiling lamp (emissive, two
// triangles, the model's light set), red left wall, green right wall, white
// floor/ceiling/back, optional front.  Flat walls carry zero vertex normals so
// the geometric normal is used.  Colours as RaySceneBuilder::BuildCornellBox.
MeshData CornellBox(glm::vec3 pos, float boxLen, float lightLen, float emission, bool front = false,
                    glm::vec3 wallAlbedo = glm::vec3(1.0f));

// Closed tube around a (2,3) torus knot with a seeded radial displacement:
// nu x nv quads = 2*nu*nv triangles, nu*nv vertices, smooth vertex normals.
// nu=2048, nv=213 gives 872,448 triangles / 436,224 vertices ("dragon class").
MeshData KnotTube(int nu, int nv, glm::vec3 center, float scale, u32 seed, const Material &material);

// Right triangular prism (8 triangles), axis along z, for the dispersion scene.
MeshData Prism(glm::vec3 center, float side, float depth, const Material &material);

// Axis-aligned emissive rectangle (two triangles) as a light-only model.
MeshData LightPanel(glm::vec3 q, glm::vec3 u, glm::vec3 v, glm::vec3 colour, float strength);

// Checker/gradient RGBA texture whose texels are k/255.0f like an 8-bit image.
MeshData::Texture CheckerTexture(int width, int height, int channels, u32 seed, int wrapS = 10497, int wrapT = 10497);

// Textured room: walls/floor with base-colour, normal, metallic-roughness,
// emissive and occlusion maps plus an alpha-cut foliage card; exercises every
// texture branch of the material resolver.
MeshData TexturedRoom(glm::vec3 pos, float boxLen, int texSize, u32 seed);

// Random triangle soup inside the box [pos, pos+boxLen]^3 for differential testing: `nTris`
// triangles of mixed sizes (a few degenerate), vertex normals that are zero, unit or skewed,
// `nMaterials` materials with every field drawn at random (including the corners: roughness 0,
// ior 1, full transmission, alpha cut, emission) and small random textures on some of them.
// The first `nLights` triangles are emissive and form the model's light set.
MeshData RandomSoup(glm::vec3 pos, float boxLen, int nTris, int nMaterials, int nLights, u32 seed);

// Distance ties on purpose: `copies` copies of the same grid of `n` x `n` squares (two triangles each)
// in the plane y = pos.y, every copy with its own colour, shuffled so that the copies of one square
// are far apart in the identifier list.  Whatever a ray hits, it hits `copies` triangles of one model
// at exactly the same distance; which one a renderer reports depends on the order it visits them in.
MeshData CoplanarStack(glm::vec3 pos, float edge, int n, int copies, u32 seed);

}  // namespace procedural
