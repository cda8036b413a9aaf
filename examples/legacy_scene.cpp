// examples/legacy_scene.cpp — a scene script of the reference's legacy design, written against the host mirror exactly
// as main.cpp:725-833 writes its own (Scene::instances of Instance(Geometry*, Material*, transform), a Rectangle light in
// both lists, one TriangleMesh placed twice), rendered on the GPU through Renderer::render_image.
//   g++ -std=c++17 examples/legacy_scene.cpp -Iraytracing_amd/host -Lraytracing_amd -lrt_host -lrt_hip -Wl,-rpath,$PWD/raytracing_amd
//   ./a.out out.bmp [width height spp]
#include "Scene.h"

#include <cstdio>
#include <cstdlib>

using namespace rt;

static Mesh tetrahedron() {
    Mesh m;
    const vec3 p[4] = {vec3(0, 1, 0), vec3(-1, -0.5f, 0.8f), vec3(1, -0.5f, 0.8f), vec3(0, -0.5f, -1)};
    const int f[4][3] = {{0, 1, 2}, {0, 2, 3}, {0, 3, 1}, {1, 3, 2}};
    for (int t = 0; t < 4; t++) {
        const vec3 n = normalize(cross(p[f[t][1]] - p[f[t][0]], p[f[t][2]] - p[f[t][0]]));
        MeshTriangle tri;
        for (int c = 0; c < 3; c++) {
            Vertex v;
            v.position = p[f[t][c]];
            v.normal = n;
            v.uv = vec2(0, 0);
            tri.indicies[c] = (uint32_t)m.verts.size();
            m.verts.push_back(v);
        }
        m.triangles.push_back(tri);
    }
    return m;
}

int main(int argc, char** argv) {
    const char* out = argc > 1 ? argv[1] : "legacy_scene.bmp";
    Options options;
    options.width = argc > 3 ? std::atoi(argv[2]) : 256;
    options.height = argc > 3 ? std::atoi(argv[3]) : 192;
    options.samples_per_pixel = argc > 4 ? std::atoi(argv[4]) : 32;
    options.max_depth = 5;
    options.output_name = out;

    Scene world;
    world.instances.push_back(Instance(new Rectangle(vec3(0, 1, 0), 3, 3), new DisneyDiffuse(vec3(0.7f), 0.4f, 0.5f), vec3(0)));
    DisneyUber* uber = new DisneyUber;
    uber->base_color = vec3(0.6f, 0.05f, 0.05f);
    uber->specular = 1.0f;
    uber->roughness = 0.1f;
    world.instances.push_back(Instance(new Sphere(0.1f), uber, vec3(0, 0.1f, 0)));
    world.instances.push_back(Instance(new Sphere(0.1f), new DisneyGlass(vec3(1), 0.01f, 0, 1.5f), vec3(0, 0.1f, 0.25f)));
    world.instances.push_back(Instance(new Box(vec3(0.2f, 0.3f, 0.15f)), new MatteMaterial(new ConstantTexture(vec3(0.2f, 0.4f, 0.8f))),
                                       translate(mat4(1.f), vec3(0.45f, 0.15f, 0.3f))));
    world.instances.push_back(Instance(new Cylinder(0.06f, 0.3f), new Microfacet(nullptr, 0.4f, 0), vec3(0.45f, 0.15f, -0.3f)));
    // a light is an Instance in BOTH lists (main.cpp:801-813)
    EmissiveMaterial* lamp = new EmissiveMaterial(vec3(17, 12, 4));
    world.instances.push_back(Instance(new Rectangle(vec3(0, -1, 0), 0.3f, 1.3f), lamp, vec3(0, 0.9f, 0)));
    world.lights.push_back(Instance(new Rectangle(vec3(0, -1, 0), 0.3f, 1.3f), lamp, vec3(0, 0.9f, 0)));
    // one mesh, its BVH built once, placed twice
    const Mesh mesh = tetrahedron();
    TriangleMesh* tm = new TriangleMesh(&mesh);
    DisneyMetal* metal = new DisneyMetal(vec3(0.7f), 0.2f, 0.0f);
    world.instances.push_back(Instance(tm, metal, scale(translate(mat4(1.f), vec3(-0.4f, 0.12f, 0.3f)), vec3(0.12f))));
    world.instances.push_back(Instance(tm, metal, scale(translate(mat4(1.f), vec3(-0.4f, 0.2f, -0.35f)), vec3(0.2f))));
    world.set_environment(make_uniform_environment(16, 8, vec3(0.05f)));
    world.build_top_level_bvh(); // Scene.cpp:6-15: the TLAS; + flatten and upload

    Camera cam(look_at(vec3(-1.4f, 0.7f, 0), vec3(0, 0.2f, 0), vec3(0, 1, 0)), 45, options.width, options.height);
    Intersection si;
    const Ray probe = cam.get_ray(options.width * 0.5f, options.height * 0.5f);
    if (world.closest_hit(probe, &si)) std::printf("centre ray: instance %d at t = %g\n", si.index, si.t);
    Renderer render;
    render.render_image(cam, world, options);
    std::printf("%llu paths, %llu rays\n", (unsigned long long)render.last_stats.paths,
                (unsigned long long)(render.last_stats.rays_closest + render.last_stats.rays_shadow));
    return render.last_stats.paths == (unsigned long long)options.width * options.height * options.samples_per_pixel ? 0 : 1;
}
