// Material.h — host mirror of the reference's material classes
// (Material.h:43-284).  Same class names and constructor arguments; on this
// side a material only HOLDS its parameters — evaluation (Eval / Sample_Eval /
// PDF, Material.cpp) runs in the HIP shade kernel, which switches on the tag
// produced by flatten().
#pragma once
#include "../../include/rt_hip.h"
#include "rt_math.h"

namespace rt {

class Texture { // Material.h:7-12
public:
    virtual ~Texture() {}
    virtual vec3 sample(float u, float v, vec3 point) const = 0;
    // c0 / c1 / s0 of an RT_MAT_MATTE record (include/rt_hip.h): the shade kernel samples the texture per hit
    virtual void describe(rt_material* m) const = 0;
};
class ConstantTexture : public Texture { // Material.h:13-24
public:
    explicit ConstantTexture(vec3 c) : color(c) {}
    ConstantTexture(float r, float g, float b) : color(r, g, b) {}
    explicit ConstantTexture(float rgb) : color(rgb) {}
    vec3 sample(float, float, vec3) const override { return color; }
    void describe(rt_material* m) const override {
        m->c0[0] = color.x;
        m->c0[1] = color.y;
        m->c0[2] = color.z;
    }

private:
    vec3 color;
};

// Material.h:25-41.  `grid` is stored as PI / grid and the cell test compares modulo(u, grid) with half of it, as the
// reference does.  (Triangle-mesh hits carry no uv — Object.h:350-355 never writes si.u / si.v — so on the mesh path
// a checkered albedo evaluates at u = v = 0: its `even` colour; a Rectangle instance of a two-level scene supplies its
// own si->u, si->v, Object.h:116-117.)
class CheckeredTexture : public Texture {
public:
    CheckeredTexture(vec3 even_, vec3 odd_, float grid_ = 0.25f) : even(even_), odd(odd_), grid(3.14159265358979323846f / grid_) {}
    static CheckeredTexture* from_record(vec3 even_, vec3 odd_, float stored_grid) { // the value describe() wrote, bit for bit
        CheckeredTexture* t = new CheckeredTexture(even_, odd_);
        t->grid = stored_grid;
        return t;
    }
    vec3 sample(float u, float v, vec3) const override {
        auto modulo = [](float a, float b) { // Math.h:39-42
            float r = std::fmod(a, b);
            return (r < 0) ? r + b : r;
        };
        const bool odd_cell = (modulo(u, grid) < grid * 0.5) ^ (modulo(v, grid) < grid * 0.5);
        return odd_cell ? odd : even;
    }
    void describe(rt_material* m) const override {
        m->c0[0] = even.x; m->c0[1] = even.y; m->c0[2] = even.z;
        m->c1[0] = odd.x; m->c1[1] = odd.y; m->c1[2] = odd.z;
        m->s[0] = grid;
    }

private:
    vec3 even, odd;
    float grid;
};

class Material { // Material.h:43-63
public:
    virtual ~Material() {}
    virtual vec3 emitted() const { return vec3(0); }
    // Tag + parameters for the GPU material table (rt_material, include/rt_hip.h).
    virtual rt_material flatten() const = 0;

protected:
    static rt_material blank(int type) {
        rt_material m{};
        m.type = type;
        return m;
    }
    static void put(float* dst, const vec3& v) {
        dst[0] = v.x;
        dst[1] = v.y;
        dst[2] = v.z;
    }
};

class MatteMaterial : public Material { // Material.h:64-92
public:
    explicit MatteMaterial(Texture* albedo_) : albedo(albedo_) {}
    ~MatteMaterial() override { delete albedo; }
    rt_material flatten() const override {
        rt_material m = blank(RT_MAT_MATTE);
        albedo->describe(&m); // sampled per hit on the GPU: a Rectangle instance supplies (u, v), every other hit (0, 0)
        return m;
    }

private:
    Texture* albedo;
};

class Microfacet : public Material { // Material.h:95-109, GGX
public:
    Microfacet(Texture* albedo_, float roughness_, float metalness_) : albedo(albedo_), roughness(roughness_), metalness(metalness_) {}
    ~Microfacet() override { delete albedo; }
    rt_material flatten() const override {
        rt_material m = blank(RT_MAT_MICROFACET_GGX);
        m.s[0] = roughness; // albedo/metalness are unused by the reference's arithmetic (Material.cpp:68-98)
        m.s[1] = metalness;
        return m;
    }

protected:
    Texture* albedo;
    float roughness, metalness;
};

// New: same lobe structure as Microfacet with a Beckmann normal distribution.
class BeckmannMicrofacet : public Microfacet {
public:
    using Microfacet::Microfacet;
    rt_material flatten() const override {
        rt_material m = Microfacet::flatten();
        m.type = RT_MAT_MICROFACET_BECKMANN;
        return m;
    }
};

class EmissiveMaterial : public Material { // Material.h:274-284
public:
    explicit EmissiveMaterial(vec3 emissive_color) : emit(emissive_color) {}
    vec3 emitted() const override { return emit; }
    rt_material flatten() const override {
        rt_material m = blank(RT_MAT_EMISSIVE);
        put(m.c0, emit);
        return m;
    }

private:
    vec3 emit;
};

class DisneyDiffuse : public Material { // Material.h:111-125
public:
    DisneyDiffuse(vec3 albedo_, float roughness_, float subsurface_) : albedo(albedo_), roughness(roughness_), subsurface(subsurface_) {}
    rt_material flatten() const override {
        rt_material m = blank(RT_MAT_DISNEY_DIFFUSE);
        put(m.c0, albedo);
        m.s[0] = roughness;
        m.s[1] = subsurface;
        return m;
    }

private:
    vec3 albedo;
    float roughness, subsurface;
};

class DisneyMetal : public Material { // Material.h:126-140
public:
    DisneyMetal(vec3 albedo_, float roughness_, float anisotropic_) : albedo(albedo_), roughness(roughness_), anisotropic(anisotropic_) {}
    rt_material flatten() const override {
        rt_material m = blank(RT_MAT_DISNEY_METAL);
        put(m.c0, albedo);
        m.s[0] = roughness;
        m.s[1] = anisotropic;
        return m;
    }

private:
    vec3 albedo;
    float roughness, anisotropic;
};

class DisneyClearcoat : public Material { // Material.h:142-154
public:
    explicit DisneyClearcoat(float clearcoat_gloss_) : clearcoat_gloss(clearcoat_gloss_) {}
    rt_material flatten() const override {
        rt_material m = blank(RT_MAT_DISNEY_CLEARCOAT);
        m.s[0] = clearcoat_gloss;
        return m;
    }

private:
    float clearcoat_gloss;
};

class DisneySheen : public Material { // Material.h:156-168
public:
    DisneySheen(vec3 base, float sheen_tint_) : base_color(base), sheen_tint(sheen_tint_) {}
    rt_material flatten() const override {
        rt_material m = blank(RT_MAT_DISNEY_SHEEN);
        put(m.c0, base_color);
        m.s[0] = sheen_tint;
        return m;
    }

private:
    vec3 base_color;
    float sheen_tint;
};

class DisneyUber : public Material { // Material.h:170-197 (public fields, as in the reference)
public:
    rt_material flatten() const override {
        rt_material m = blank(RT_MAT_DISNEY_UBER);
        put(m.c0, base_color);
        const float s[12] = {specular_transmission, metallic, subsurface, specular, roughness, specular_tint,
                             anisotropic, sheen, sheen_tint, clearcoat, clearcoat_gloss, ior};
        for (int i = 0; i < 12; i++) m.s[i] = s[i];
        return m;
    }
    vec3 base_color = vec3(0);
    float specular_transmission = 0, metallic = 0, subsurface = 0, specular = 0, roughness = 0, specular_tint = 0;
    float anisotropic = 0, sheen = 0, sheen_tint = 0, clearcoat = 0, clearcoat_gloss = 0;
    float ior = 1.0;
};

class RoughDielectric : public Material { // Material.h:199-219
public:
    RoughDielectric(float eta, float roughness_, vec3 specular_reflectance_, vec3 specular_transmittance_)
        : eta_i_e(eta), roughness(roughness_), specular_reflectance(specular_reflectance_), specular_transmittance(specular_transmittance_) {}
    rt_material flatten() const override {
        rt_material m = blank(RT_MAT_ROUGH_DIELECTRIC);
        put(m.c0, specular_reflectance);
        put(m.c1, specular_transmittance);
        m.s[0] = eta_i_e;
        m.s[1] = roughness;
        return m;
    }

private:
    float eta_i_e, roughness;
    vec3 specular_reflectance, specular_transmittance;
};

class DisneyGlass : public Material { // Material.h:221-237
public:
    DisneyGlass(vec3 albedo_, float roughness_, float anisotropic_, float eta) : albedo(albedo_), roughness(roughness_), anisotropic(anisotropic_), eta_ie(eta) {}
    rt_material flatten() const override {
        rt_material m = blank(RT_MAT_DISNEY_GLASS);
        put(m.c0, albedo);
        m.s[0] = roughness;
        m.s[1] = anisotropic;
        m.s[2] = eta_ie;
        return m;
    }

private:
    vec3 albedo;
    float roughness, anisotropic, eta_ie;
};

class GlassMaterial : public Material { // Material.h:240-273
public:
    explicit GlassMaterial(float index_refraction) : index_r(index_refraction) {}
    rt_material flatten() const override {
        rt_material m = blank(RT_MAT_GLASS);
        m.s[0] = index_r;
        return m;
    }

private:
    float index_r;
};

// Re-creates a Material object from a flattened entry (used by the C API so
// that scripted scenes can hand over plain parameter records).
Material* material_from_record(const rt_material& rec);

} // namespace rt
