/*
 * ref_tinyobj.cpp — drives the reference's VENDORED tinyobjloader v1.0.6
 * ($(REF)/libs/tiny_obj_loader/tiny_obj_loader.h, compiled in place) the way
 * common/loader.hpp:11-66 does, and dumps the resulting triangle array (60-byte records).
 * TEST INFRASTRUCTURE: validates the product's own OBJ readers (scenes.load_obj,
 * app/restir_main.cpp) — triangle ORDER defines primID and light order. loader.hpp itself
 * cannot be included (it pulls Orochi, an empty submodule).
 *   ref_tinyobj scene.obj mtl_basedir out.tris
 */
#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"

#include <cstdio>
#include <string>
#include <vector>

struct Tri { float v[3][3]; float color[3]; float emissive[3]; };

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: ref_tinyobj scene.obj mtl_basedir out.tris\n"); return 2; }
    tinyobj::attrib_t attrib;
    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string err;
    tinyobj::LoadObj(&attrib, &shapes, &materials, &err, argv[1], argv[2]);
    std::vector<Tri> out;
    for (size_t s = 0; s < shapes.size(); s++)
    {
        size_t index_offset = 0;
        for (size_t f = 0; f < shapes[s].mesh.num_face_vertices.size(); f++)
        {
            const size_t fv = size_t(shapes[s].mesh.num_face_vertices[f]);
            Tri t;
            for (size_t v = 0; v < fv && v < 3; v++)
            {
                const tinyobj::index_t idx = shapes[s].mesh.indices[index_offset + v];
                for (int k = 0; k < 3; ++k) t.v[v][k] = attrib.vertices[3 * size_t(idx.vertex_index) + k];
            }
            index_offset += fv;
            const int m = shapes[s].mesh.material_ids[f];
            for (int k = 0; k < 3; ++k)
            {
                t.color[k] = m >= 0 ? materials[m].diffuse[k] : 0.0f;
                t.emissive[k] = m >= 0 ? materials[m].emission[k] : 0.0f;
            }
            out.push_back(t);
        }
    }
    FILE* fo = fopen(argv[3], "wb");
    fwrite(out.data(), sizeof(Tri), out.size(), fo);
    fclose(fo);
    printf("%zu\n", out.size());
    return 0;
}
