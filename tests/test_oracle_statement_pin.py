"""Build-container only (skipped where /root/reference is absent): a MECHANICAL pin of the oracle's restatement of the
reference's device programs. The reference has no golden images and its .cu files need <optix.h>, so the shading half of
the oracle cannot be executed against the reference (Oracle). What can be checked is that every function the
oracle restates says, statement for statement, what the reference's function says: both texts are stripped of comments,
preprocessor branches (config.h values) and qualifiers, the DOCUMENTED renames below are applied, and the statement
lists must then be equal. The reference is read as text at test time; nothing of it is stored in the repository.

Renames (oracle name <- reference name), the only liberties the restatement takes:
  pm_sinf / pm_cosf / pm_expf / pm_atanf / pm_atan2f / pm_acosf <- sinf / cosf / expf / atanf / atan2f / acosf
      (portable single-precision functions, oracle/orc_math.h: the reference built with --use_fast_math, so no libm is
      "the" definition; tests/test_oracle_math.py bounds the distance)
  expf3 <- expf on float3, fmaxf3 <- fmaxf(float3)          (vector_math.h overloads spelled out)
  M_PIf_ / M_1_PIf_ <- M_PIf / M_1_PIf                     (constants of vector_math.h under names libm does not claim)
  copysignf <- copysign; clampi <- clamp on int; std::min / std::max <- min / max
  sysData textures: tex2D(sysData.textures[k], ..) <- tex2D<float4>(sysData.envTexture | material.texture*, ..)
  lens / light / BSDF callables: plain functions <- __direct_callable__*; the camera comes from sysData in both
OptiX-specific statements (intrinsics, optixTrace, optixDirectCall, SBT data) have no C++ counterpart and are mapped one to
one by the table OPTIX_STATEMENTS below; every other statement must match literally.
"""
import os
import re

import pytest

REF = "/root/reference/apps/rtigo3/shaders"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference sources are only present in the build container")

def _diff(ref, orc):
    import difflib
    return "\n".join(difflib.unified_diff(ref, orc, "reference", "oracle", lineterm="", n=1))[:6000]



DEFINES = {"USE_NEXT_EVENT_ESTIMATION": 1, "USE_DEBUG_EXCEPTIONS": 0, "USE_TIME_VIEW": 0}  # shaders/config.h:52-60


def preprocess(text):
    """Comments out, #if NAME / #else / #endif resolved with DEFINES, other # lines dropped."""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out, stack = [], []
    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("#"):
            m = re.match(r"#\s*if\s+(\w+)", s)
            if m:
                stack.append(bool(DEFINES.get(m.group(1), 0)))
            elif re.match(r"#\s*(ifdef|ifndef)", s):
                stack.append(True)
            elif re.match(r"#\s*else", s) and stack:
                stack[-1] = not stack[-1]
            elif re.match(r"#\s*endif", s) and stack:
                stack.pop()
            continue
        if all(stack):
            out.append(line)
    return "\n".join(out)


def function_body(text, name):
    """Text between the braces of the definition of `name` (first definition in `text`)."""
    for m in re.finditer(r"\b" + re.escape(name) + r"\s*\(", text):
        depth, i = 0, m.end() - 1
        while i < len(text):  # skip the parameter list
            depth += text[i] == "("
            depth -= text[i] == ")"
            i += 1
            if depth == 0:
                break
        j = i
        while j < len(text) and text[j] in " \t\r\n":
            j += 1
        if j >= len(text) or text[j] != "{":
            continue  # a call or a declaration, not the definition
        depth, k = 0, j
        while k < len(text):
            depth += text[k] == "{"
            depth -= text[k] == "}"
            k += 1
            if depth == 0:
                return text[j + 1:k - 1]
    raise AssertionError(f"definition of {name} not found")


REF_RENAMES = [
    (r"\bsinf\b", "pm_sinf"), (r"\bcosf\b", "pm_cosf"), (r"\batan2f\b", "pm_atan2f"), (r"\batanf\b", "pm_atanf"), (r"\bacosf\b", "pm_acosf"),
    (r"\bM_PIf\b", "M_PIf_"), (r"\bM_1_PIf\b", "M_1_PIf_"), (r"\bcopysign\b", "copysignf"),
    (r"tex2D<float4>\(sysData\.envTexture,", "tex2D(sysData.textures[2],"),
    (r"tex2D<float4>\(material\.textureAlbedo,", "tex2D(sysData.textures[0],"),
    (r"tex2D<float4>\(material\.textureCutout,", "tex2D(sysData.textures[1],"),
    (r"\bexpf\(-", "expf3(-"),                      # the two float3 overload uses: absorption along a segment
    (r"\bfmaxf\(throughput\)", "fmaxf3(throughput)"),
    (r"\bclamp\(static_cast<int>", "clampi(static_cast<int>"),
    (r"(?<![\w:])min\(", "std::min("), (r"(?<![\w:])max\(", "std::max("),
]
ORC_DROPS = [r"\(void\)\s*\w+\s*;"]  # unused-parameter markers


SPELLING = [  # benign spellings, normalised away on BOTH sides
    (r"float\(sysData\.numLights\)", "sysData.numLights"),   # int -> float conversion, explicit or through vector_math.h's overloads
    (r"\.data\(\)", ""),                                     # std::vector instead of a device pointer
    (r"\(size_t\)\s*", ""), (r"\(int\)\s*(?=MATERIAL_STACK_)", ""),
]


def _split_control(stmt):
    """'if(c)x=1;' -> ['if(c)', 'x=1;'] (also else / while / for), so that brace style does not matter."""
    out = []
    while True:
        m = re.match(r"(else\b)\s*", stmt)
        if m:
            out.append("else")
            stmt = stmt[m.end():]
            continue
        m = re.match(r"(if|while|for)\s*\(", stmt)
        if not m:
            break
        depth, i = 0, m.end() - 1
        while i < len(stmt):
            depth += stmt[i] == "("
            depth -= stmt[i] == ")"
            i += 1
            if depth == 0:
                break
        out.append(stmt[:i])
        stmt = stmt[i:].strip()
    if stmt:
        out.append(stmt)
    return out


def statements(body, renames=(), oracle=False):
    for pat, rep in renames:
        body = re.sub(pat, rep, body)
    if oracle:
        for pat in ORC_DROPS:
            body = re.sub(pat, "", body)
    for pat, rep in SPELLING:
        body = re.sub(pat, rep, body)
    body = re.sub(r"\s+", " ", body)
    body = re.sub(r"\s*([(),*&<>=+\-/?:!|\[\]])\s*", r"\1", body)
    out = []
    for part in re.split(r"[{}]", body):
        # split at ';' outside parentheses (for-loop headers keep theirs)
        depth, cur = 0, ""
        for ch in part:
            depth += ch == "("
            depth -= ch == ")"
            cur += ch
            if ch == ";" and depth == 0:
                out.extend(_split_control(cur.strip()))
                cur = ""
        if cur.strip():
            out.extend(_split_control(cur.strip()))
    return [x for x in out if x and x != ";"]


def read(path):
    with open(path) as f:
        return preprocess(f.read())


def load_texts():
    t = {name: read(os.path.join(REF, name)) for name in
         ("bxdf_diffuse.cu", "bxdf_specular.cu", "bxdf_ggx_smith.cu", "light_sample.cu", "miss.cu", "lens_shader.cu", "closesthit.cu",
          "raygeneration.cu", "shader_common.h", "random_number_generators.h", "anyhit.cu")}
    t["orc_shaders"] = read(os.path.join(ROOT, "oracle", "orc_shaders.h"))
    t["orc_render"] = read(os.path.join(ROOT, "oracle", "orc_render.cpp"))
    return t


@pytest.fixture(scope="module")
def texts():
    return load_texts()


class flipped:
    """`with flipped(USE_NEXT_EVENT_ESTIMATION=0):` — both texts are preprocessed with that config.h value (the oracle carries the
    reference's #if blocks and is compiled once per value: oracle/Makefile liboracle_nee0.so, liboracle_dbgexc.so)."""

    def __init__(self, **values):
        self.values = values

    def __enter__(self):
        self.saved = dict(DEFINES)
        DEFINES.update(self.values)

    def __exit__(self, *exc):
        DEFINES.clear()
        DEFINES.update(self.saved)


# (reference file, reference function, oracle file, oracle function)
PLAIN = [
    ("random_number_generators.h", "tea", "orc_shaders", "tea"),
    ("random_number_generators.h", "rng", "orc_shaders", "rng"),
    ("random_number_generators.h", "rng2", "orc_shaders", "rng2"),
    ("shader_common.h", "refract", "orc_shaders", "refract"),
    ("shader_common.h", "powerHeuristic", "orc_shaders", "powerHeuristic"),
    ("shader_common.h", "intensity", "orc_shaders", "intensity"),
    ("bxdf_diffuse.cu", "alignVector", "orc_shaders", "alignVector"),
    ("bxdf_diffuse.cu", "unitSquareToCosineHemisphere", "orc_shaders", "unitSquareToCosineHemisphere"),
    ("bxdf_diffuse.cu", "__direct_callable__sample_brdf_diffuse", "orc_shaders", "sample_brdf_diffuse"),
    ("bxdf_diffuse.cu", "__direct_callable__eval_brdf_diffuse", "orc_shaders", "eval_brdf_diffuse"),
    ("bxdf_specular.cu", "evaluateFresnelDielectric", "orc_shaders", "evaluateFresnelDielectric"),
    ("bxdf_specular.cu", "__direct_callable__sample_brdf_specular", "orc_shaders", "sample_brdf_specular"),
    ("bxdf_specular.cu", "__direct_callable__sample_bsdf_specular", "orc_shaders", "sample_bsdf_specular"),
    ("bxdf_ggx_smith.cu", "distribution_d_pdf", "orc_shaders", "distribution_d_pdf"),
    ("bxdf_ggx_smith.cu", "distribution_sample", "orc_shaders", "distribution_sample"),
    ("bxdf_ggx_smith.cu", "smith_G1", "orc_shaders", "smith_G1"),
    ("bxdf_ggx_smith.cu", "distribution_G", "orc_shaders", "distribution_G"),
    ("bxdf_ggx_smith.cu", "__direct_callable__sample_brdf_ggx_smith", "orc_shaders", "sample_brdf_ggx_smith"),
    ("bxdf_ggx_smith.cu", "__direct_callable__eval_brdf_ggx_smith", "orc_shaders", "eval_brdf_ggx_smith"),
    ("bxdf_ggx_smith.cu", "__direct_callable__sample_bsdf_ggx_smith", "orc_shaders", "sample_bsdf_ggx_smith"),
    ("light_sample.cu", "unitSquareToSphere", "orc_shaders", "unitSquareToSphere"),
    ("light_sample.cu", "__direct_callable__light_env_constant", "orc_shaders", "light_env_constant"),
    ("light_sample.cu", "__direct_callable__light_env_sphere", "orc_shaders", "light_env_sphere"),
    ("light_sample.cu", "__direct_callable__light_parallelogram", "orc_shaders", "light_parallelogram"),
    ("lens_shader.cu", "__direct_callable__pinhole", "orc_shaders", "lens_pinhole"),
    ("lens_shader.cu", "__direct_callable__fisheye", "orc_shaders", "lens_fisheye"),
    ("lens_shader.cu", "__direct_callable__sphere", "orc_shaders", "lens_sphere"),
]


@pytest.mark.parametrize("ref_file,ref_fn,orc_file,orc_fn", PLAIN, ids=[p[3] for p in PLAIN])
def test_callable_says_what_the_reference_says(texts, ref_file, ref_fn, orc_file, orc_fn):
    ref = statements(function_body(texts[ref_file], ref_fn), REF_RENAMES)
    orc = statements(function_body(texts[orc_file], orc_fn), oracle=True)
    assert orc == ref, _diff(ref, orc)


# ---- programs that talk to OptiX: intrinsics and traces have no C++ counterpart; one-to-one statement map -----------
# reference statement (after the renames above) -> oracle statement. Everything not listed must match literally.
OPTIX_STATEMENTS = {
    # closesthit.cu: SBT data, primitive index, barycentrics, transforms, payload, ray tmax
    "GeometryInstanceData*theData=reinterpret_cast<GeometryInstanceData*>(optixGetSbtDataPointer());": "const Geometry&g=*hc.geom;",
    "const unsigned int thePrimitiveIndex=optixGetPrimitiveIndex();": None,
    "const uint3*indices=reinterpret_cast<uint3*>(theData->indices);": None,
    "const TriangleAttributes*attributes=reinterpret_cast<TriangleAttributes*>(theData->attributes);": None,
    "const uint3 tri=indices[thePrimitiveIndex];": "const unsigned int*tri=&g.indices[3*hc.primitive];",
    "TriangleAttributes const&attr0=attributes[tri.x];": "TriangleAttributes const&attr0=g.attributes[tri[0]];",
    "TriangleAttributes const&attr1=attributes[tri.y];": "TriangleAttributes const&attr1=g.attributes[tri[1]];",
    "TriangleAttributes const&attr2=attributes[tri.z];": "TriangleAttributes const&attr2=g.attributes[tri[2]];",
    "const float2 theBarycentrics=optixGetTriangleBarycentrics();": "const float2 theBarycentrics=make_float2(hc.beta,hc.gamma);",
    "float4 objectToWorld[3];": "const float*objectToWorld=hc.inst->objectToWorld;",
    "float4 worldToObject[3];": "const float*worldToObject=hc.inst->worldToObject;",
    "getTransforms(objectToWorld,worldToObject);": None,
    "state.tangent=normalize(transformVector(objectToWorld,tg));": "state.tangent=normalize(xfmVector(objectToWorld,tg));",
    "PerRayData*thePrd=mergePointer(optixGetPayload_0(),optixGetPayload_1());": None,
    "thePrd->distance=optixGetRayTmax();": "thePrd->distance=hc.tmax;",
    "if(0<=theData->lightIndex&&(thePrd->flags&FLAG_FRONTFACE))": "if(0<=hc.inst->light&&(thePrd->flags&FLAG_FRONTFACE))",
    "LightDefinition const&light=sysData.lightDefinitions[theData->lightIndex];": "LightDefinition const&light=sysData.lightDefinitions[hc.inst->light];",
    "MaterialDefinition const&material=sysData.materialDefinitions[theData->materialIndex];": "MaterialDefinition const&material=sysData.materialDefinitions[hc.inst->material];",
    # closesthit.cu: callables and the shadow ray
    "const int indexBSDF=NUM_LENS_SHADERS+NUM_LIGHT_TYPES+material.indexBSDF*2;": None,
    "optixDirectCall<void,MaterialDefinition const&,State const&,PerRayData*>(indexBSDF,material,state,thePrd);": "callBsdfSample(material.indexBSDF,material,state,thePrd);",
    "const int indexLightType=NUM_LENS_SHADERS+sysData.lightDefinitions[lightSample.index].type;": None,
    "optixDirectCall<void,float3 const&,const float2,LightSample&>(indexLightType,thePrd->pos,sample,lightSample);": "callLight(sysData,sysData.lightDefinitions[lightSample.index].type,thePrd->pos,sample,lightSample);",
    "const float4 bsdf_pdf=optixDirectCall<float4,MaterialDefinition const&,State const&,PerRayData*,float3 const&>(indexBSDF+1,material,state,thePrd,lightSample.direction);": "const float4 bsdf_pdf=callBsdfEval(material.indexBSDF,material,state,thePrd,lightSample.direction);",
    "unsigned int p0=optixGetPayload_0();": None,
    "unsigned int p1=optixGetPayload_1();": None,
    "optixTrace(sysData.topObject,thePrd->pos,lightSample.direction,sysData.sceneEpsilon,lightSample.distance-sysData.sceneEpsilon,0.0f,OptixVisibilityMask(0xFF),OPTIX_RAY_FLAG_DISABLE_CLOSESTHIT,RAYTYPE_SHADOW,NUM_RAYTYPES,RAYTYPE_SHADOW,p0,p1);":
        "const bool shadowed=traceShadow(o,thePrd,thePrd->pos,lightSample.direction,sysData.sceneEpsilon,lightSample.distance-sysData.sceneEpsilon); if(shadowed)thePrd->flags|=FLAG_SHADOW;",
    # miss.cu
    # raygeneration.cu integrator
    "uint2 payload=splitPointer(&prd);": None,
}


def apply_optix_map(ref_statements):
    out = []
    for s in ref_statements:
        if s in OPTIX_STATEMENTS:
            rep = OPTIX_STATEMENTS[s]
            if rep is not None:
                for x in re.split(r"(?<=;) ", rep):
                    out.extend(_split_control(x.strip()))
        else:
            out.append(s)
    return out


def test_closesthit_radiance(texts):
    """__closesthit__radiance (closesthit.cu:126-305) vs closesthitRadiance: the rtigo3 rule. The Optix7Gui light rule
    (an inserted block, apps/Optix7Gui/shaders/closesthit.cu:189-226) and the denoiser albedo / normal assignments are
    additions of this build guarded by o.shaderVariant / o.aov and are cut out before the comparison."""
    ref = apply_optix_map(statements(function_body(texts["closesthit.cu"], "__closesthit__radiance"), REF_RENAMES))
    body = function_body(texts["orc_render"], "closesthitRadiance")
    body = re.sub(r"if \(0 <= hc\.inst->light && o\.shaderVariant == TWK_SHADERS_OPTIX7GUI\)\s*\{.*?\n  \}\n", "", body, flags=re.S)
    orc = statements(body, oracle=True)
    orc = [s for s in orc if s not in ("const SystemData&sysData=o.sys;", "thePrd->normal=state.normal;", "thePrd->albedo=emission;", "thePrd->albedo=state.albedo;")]
    orc = [s.replace("|=FLAG_LIGHT|FLAG_TERMINATE", "|=FLAG_TERMINATE") for s in orc]
    assert orc == ref, _diff(ref, orc)


def test_miss_programs(texts):
    for ref_fn, case in (("__miss__env_null", 0), ("__miss__env_constant", 1), ("__miss__env_sphere", 2)):
        ref = statements(function_body(texts["miss.cu"], ref_fn), REF_RENAMES)
        ref = [s for s in ref if s != "PerRayData*thePrd=mergePointer(optixGetPayload_0(),optixGetPayload_1());"]
        body = function_body(texts["orc_render"], "missProgram")
        m = re.search(r"case %d:\s*(\{)?(.*?)break;" % case, body, flags=re.S)
        orc = statements(m.group(2), oracle=True)
        orc = [s for s in orc if not s.startswith("thePrd->albedo=")]
        orc = [s.replace("|=FLAG_LIGHT|FLAG_TERMINATE", "|=FLAG_TERMINATE") for s in orc]
        assert orc == ref, f"{ref_fn}\n" + _diff(ref, orc)


def test_integrator_loop(texts):
    """integrator (raygeneration.cu:42-149): the loop body. optixTrace + the hit / miss program dispatch OptiX performs
    internally are the oracle's traceRadiance + explicit dispatch; the AOV block is this build's (Optix7Gui) addition."""
    ref = statements(function_body(texts["raygeneration.cu"], "integrator"), REF_RENAMES)
    ref = apply_optix_map(ref)
    trace = [s for s in ref if s.startswith("optixTrace(")]
    assert trace == ["optixTrace(sysData.topObject,prd.pos,prd.wi,sysData.sceneEpsilon,prd.distance,0.0f,OptixVisibilityMask(0xFF),OPTIX_RAY_FLAG_NONE,RAYTYPE_RADIANCE,NUM_RAYTYPES,RAYTYPE_RADIANCE,payload.x,payload.y);"]
    body = function_body(texts["orc_render"], "integrator")
    body = re.sub(r"if \(o\.aov\)\s*\{.*?\n    \}\n", "", body, flags=re.S)
    orc = statements(body, oracle=True)
    # the oracle's stand-in for the trace and the program dispatch, in place of the single optixTrace statement
    i = orc.index("const Hit h=traceRadiance(o,&prd,prd.pos,prd.wi,sysData.sceneEpsilon,prd.distance);")
    j = orc.index("closesthitRadiance(o,hc,&prd);") + 1
    orc = orc[:i] + trace + orc[j:]
    drop = {"const SystemData&sysData=o.sys;", "albedo=make_float3(0.0f);", "normal=make_float3(0.0f);", "prd.normal=make_float3(0.0f);"}
    orc = [s for s in orc if s not in drop]
    assert orc == ref, _diff(ref, orc)


# =====================================================================================================================
# Round 3: the programs round 2 left unpinned ("What's missing" 3): __raygen__path_tracer, distribute,
# __anyhit__*, compositor, the Optix7Gui light block and denoiser AOV code, Texture::calculateSphericalCDF +
# gaussianFilter, and the screenshot tonemap loop. Same method: statement lists must be EQUAL after the documented
# renames and the one-to-one statement maps below; what is this build's own (tiled output index, seed by absolute pixel,
# AOV additions to the rtigo3 programs) is named in the map, never silently dropped.
# =====================================================================================================================
REF7 = "/root/reference/apps/Optix7Gui/shaders"
REF_SRC = "/root/reference/apps/rtigo3/src"

CASTS = [(r"\(unsigned int\)\s*", ""), (r"reinterpret_cast<(?:const )?float4\s*\*>\(([^()]*)\)", r"\1")]  # benign on both sides


def stmts(body, renames=(), extra_spelling=()):
    for pat, rep in extra_spelling:
        body = re.sub(pat, rep, body)
    return statements(body, renames)


def mapped(ref_statements, table):
    out = []
    for s in ref_statements:
        if s in table:
            rep = table[s]
            if rep is not None:
                for x in (rep if isinstance(rep, list) else [rep]):
                    out.extend(_split_control(x))
        else:
            out.append(s)
    return out


def test_distribute(texts):
    ref = stmts(function_body(texts["raygeneration.cu"], "distribute"), [(r"launchIndex\.x", "x"), (r"launchIndex\.y", "y")])
    orc = stmts(function_body(texts["orc_render"], "distribute"))
    assert orc == ref, _diff(ref, orc)


def test_compositor():
    ref_text = read(os.path.join(REF, "compositor.cu"))
    ref = stmts(function_body(ref_text, "compositor"), extra_spelling=CASTS)
    ref = mapped(ref, {  # the launch index comes from the caller's loops instead of blockIdx / threadIdx
        "const unsigned int xLaunch=blockIdx.x*blockDim.x+threadIdx.x;": None,
        "const unsigned int yLaunch=blockIdx.y*blockDim.y+threadIdx.y;": None,
    })
    orc = stmts(function_body(read(os.path.join(ROOT, "oracle", "orc_render.cpp")), "compositor"), extra_spelling=CASTS)
    assert orc == ref, _diff(ref, orc)


ANYHIT_RENAMES = REF_RENAMES + [
    (r"attributes\[tri\.x\]", "g.attributes[tri[0]]"), (r"attributes\[tri\.y\]", "g.attributes[tri[1]]"), (r"attributes\[tri\.z\]", "g.attributes[tri[2]]"),
    (r"thePrd->seed", "seed"), (r"theData->materialIndex", "hc.inst->material"),
]
ANYHIT_MAP = {
    "GeometryInstanceData*theData=reinterpret_cast<GeometryInstanceData*>(optixGetSbtDataPointer());": ["const SystemData&sysData=o.sys;", "const Geometry&g=*hc.geom;"],
    "const uint3*indices=reinterpret_cast<uint3*>(theData->indices);": None,
    "const TriangleAttributes*attributes=reinterpret_cast<TriangleAttributes*>(theData->attributes);": None,
    "const unsigned int thePrimitiveIndex=optixGetPrimitiveIndex();": None,
    "const uint3 tri=indices[thePrimitiveIndex];": "const unsigned int*tri=&g.indices[3*hc.primitive];",
    "const float2 theBarycentrics=optixGetTriangleBarycentrics();": "const float2 theBarycentrics=make_float2(hc.beta,hc.gamma);",
    "PerRayData*thePrd=mergePointer(optixGetPayload_0(),optixGetPayload_1());": None,
    "optixIgnoreIntersection();": "return true;",   # the candidate is skipped
    "thePrd->flags|=FLAG_SHADOW;": None,              # set by the caller from the return value (closesthitRadiance: if(shadowed) ...)
    "optixTerminateRay();": "return false;",          # the candidate is accepted and ends the shadow ray
}


def test_anyhit_programs(texts):
    orc_text = texts["orc_render"]
    ref = mapped(stmts(function_body(texts["anyhit.cu"], "__anyhit__radiance_cutout"), ANYHIT_RENAMES), ANYHIT_MAP)
    orc = stmts(function_body(orc_text, "anyhitRadianceCutout"))
    assert orc[-1] == "return false;"   # falling off the end of the program = the candidate is accepted
    assert orc[:-1] == ref, _diff(ref, orc[:-1])
    ref = mapped(stmts(function_body(texts["anyhit.cu"], "__anyhit__shadow_cutout"), ANYHIT_RENAMES), ANYHIT_MAP)
    orc = stmts(function_body(orc_text, "anyhitShadowCutout"))
    assert orc == ref, _diff(ref, orc)
    # __anyhit__shadow (opaque materials): flag + terminate on the FIRST candidate = an any-hit query
    ref = stmts(function_body(texts["anyhit.cu"], "__anyhit__shadow"))
    assert ref == ["PerRayData*thePrd=mergePointer(optixGetPayload_0(),optixGetPayload_1());", "thePrd->flags|=FLAG_SHADOW;", "optixTerminateRay();"]
    shadow = stmts(function_body(orc_text, "traceShadow"))
    assert "return o.scene.trace(org,dir,tmin,tmax,true).instance>=0;" in shadow
    # and the radiance / shadow loops call the two cutout programs per candidate, closest first
    assert "if(!anyhitShadowCutout(o,hc,shadowSeed))" in shadow
    assert "if(!anyhitRadianceCutout(o,hc,prd->seed))" in stmts(function_body(orc_text, "traceRadiance"))


RAYGEN_RENAMES = REF_RENAMES + [(r"theLaunchIndex\.x", "lx"), (r"theLaunchIndex\.y", "ly"), (r"\bisnan\(", "std::isnan("), (r"\bisinf\(", "std::isinf("), (r"\bbuffer\[index\]", "o.output[index]")]
RAYGEN_MAP = {
    "const uint2 theLaunchIndex=make_uint2(optixGetLaunchIndex());": "const SystemData&sysData=o.sys;",   # lx, ly are parameters
    "unsigned int launchColumn=lx;": ["unsigned int launchColumn=lx;", "const bool tiled=(sysData.distribution&&1<sysData.deviceCount);"],
    "if(sysData.distribution&&1<sysData.deviceCount)": "if(tiled)",
    "launchColumn=distribute(theLaunchIndex);": "launchColumn=distribute(sysData,lx,ly);",
    "PerRayData prd;": ["PerRayData prd;", "memset(&prd,0,sizeof(prd));"],
    "const uint2 theLaunchDim=make_uint2(optixGetLaunchDimensions());": None,
    # THE stated deviation: seed by absolute pixel for every device count; identical for one device
    "const unsigned int seedIndex=theLaunchDim.x*ly+launchColumn*sysData.deviceCount+sysData.deviceIndex;": "const unsigned int seedIndex=sysData.resolution.x*ly+launchColumn;",
    "const float2 screen=make_float2(sysData.resolution);": "const float2 screen=make_float2(float(sysData.resolution.x),float(sysData.resolution.y));",
    "const float2 pixel=make_float2(launchColumn,ly);": "const float2 pixel=make_float2(float(launchColumn),float(ly));",
    "optixDirectCall<void,const float2,const float2,const float2,float3&,float3&>(sysData.lensShader,screen,pixel,sample,prd.pos,prd.wi);": "callLens(sysData,sysData.lensShader,screen,pixel,sample,prd.pos,prd.wi);",
    "float3 radiance=integrator(prd);": ["float3 albedo,normal;", "float3 radiance=integrator(o,prd,o.captureFirstHits?&o.firstHits[index]:nullptr,albedo,normal);", "rayTally().samples++;"],
    "float4*buffer=sysData.outputBuffer;": None,   # (after the cast normalisation) the output lives in the Oracle object
    # single buffer :229 `ly * resolution.x + launchColumn`; local copy :318 `theLaunchDim.x * ly + lx` — one expression here
    "const unsigned int index=ly*sysData.resolution.x+launchColumn;": None,
    "const float4 dst=o.output[index];": None,
    "radiance=lerp(make_float3(dst),radiance,1.0f/float(sysData.iterationIndex+1));": ["const float t=1.0f/float(sysData.iterationIndex+1);", "const float4 dst=o.output[index];", "radiance=lerp(make_float3(dst),radiance,t);"],
}


def test_raygen_path_tracer(texts):
    ref = mapped(stmts(function_body(texts["raygeneration.cu"], "__raygen__path_tracer"), RAYGEN_RENAMES, CASTS), RAYGEN_MAP)
    body = function_body(texts["orc_render"], "raygenPathTracer")
    body = re.sub(r"if \(o\.aov\)[^\n]*\n\s*\{.*?\n\s*\}\n", "", body, flags=re.S)   # Optix7Gui's AOV running means: test_optix7gui_*
    orc = stmts(body, extra_spelling=CASTS)
    index = "const unsigned int index=tiled?(ly*o.launchWidth+lx):(ly*sysData.resolution.x+launchColumn);"
    assert index in orc, orc
    orc.remove(index)                      # computed before the integrator call (first-hit capture needs it); see RAYGEN_MAP
    assert orc == ref, _diff(ref, orc)


def _read7(name):
    global DEFINES
    saved = dict(DEFINES)
    DEFINES.update({"USE_DENOISER_ALBEDO": 1, "USE_DENOISER_NORMAL": 1, "USE_FP32_OUTPUT": 1})  # app_config.h:59-67; the normal AOV is built too
    try:
        return read(os.path.join(REF7, name))
    finally:
        DEFINES.clear()
        DEFINES.update(saved)


def _between(seq, first, last):
    i = next(k for k, s in enumerate(seq) if s.startswith(first))
    j = next(k for k, s in enumerate(seq) if k >= i and s.startswith(last))
    return seq[i:j + 1]


def test_optix7gui_light_block(texts):
    """apps/Optix7Gui/shaders/closesthit.cu:189-226 vs the block closesthitRadiance runs under TWK_SHADERS_OPTIX7GUI."""
    ren = REF_RENAMES + [(r"sysParameter", "sysData"), (r"theData->lightIndex", "hc.inst->light")]
    ref = stmts(function_body(_read7("closesthit.cu"), "__closesthit__radiance"), ren)
    ref = _between(ref, "if(0<=hc.inst->light)", "return;")
    body = function_body(texts["orc_render"], "closesthitRadiance")
    orc = _between(stmts(body), "if(0<=hc.inst->light&&o.shaderVariant==TWK_SHADERS_OPTIX7GUI)", "return;")
    ref[0] = ref[0].replace("if(0<=hc.inst->light)", "if(0<=hc.inst->light&&o.shaderVariant==TWK_SHADERS_OPTIX7GUI)")
    # FLAG_HIT: Optix7Gui sets it with FLAG_FRONTFACE at :172 for every hit; here it is only needed by the normal AOV, so it is set where the path ends
    orc = [s.replace("(FLAG_HIT|FLAG_LIGHT|FLAG_TERMINATE)", "(FLAG_LIGHT|FLAG_TERMINATE)") for s in orc]
    assert orc == ref, _diff(ref, orc)


def test_optix7gui_aov_code(texts):
    """Denoiser AOVs: albedo / normal capture in the integrator loop (apps/Optix7Gui/shaders/raygeneration.cu:125-164) and
    their running means in the raygen program (:239-262)."""
    ren = REF_RENAMES + [(r"sysParameter\.cameraU", "cam.U"), (r"sysParameter\.cameraV", "cam.V"), (r"sysParameter\.cameraW", "cam.W"), (r"sysParameter", "sysData"),
                         (r"bufferRGBA\[index\]", "o.output[index]"), (r"bufferAlbedo\[index\]", "o.aovAlbedo[index]"), (r"bufferNormal\[index\]", "o.aovNormal[index]"),
                         (r"\bisnan\(", "std::isnan(")]
    ref = stmts(function_body(_read7("raygeneration.cu"), "__raygen__pathtracer"), ren)
    integ = stmts(function_body(texts["orc_render"], "integrator"))
    # capture block
    r = _between(ref, "if(!(prd.flags&FLAG_ALBEDO)", "normal=make_float3(dot(prd.normal")
    o = _between(integ, "if(!(prd.flags&FLAG_ALBEDO)", "normal=make_float3(dot(prd.normal")
    r = mapped(r, {
        # vector_math.h:562-565 clamp(float3, a, b) written out per component
        "albedo=clamp(throughput*prd.albedo,0.0f,1.0f);": ["const float3 a=throughput*prd.albedo;", "albedo=make_float3(clampf(a.x,0.0f,1.0f),clampf(a.y,0.0f,1.0f),clampf(a.z,0.0f,1.0f));"],
        # FLAG_HIT of Optix7Gui's closest hit (:172) = the radiance ray hit something
        "if(depth==0&&(prd.flags&FLAG_HIT))": ["if(depth==0&&h.instance>=0)", "const CameraDefinition&cam=sysData.cameraDefinitions[0];"],
    })
    assert o == r, _diff(r, o)
    # running means
    r = _between(ref, "if(!(std::isnan(radiance.x)", "o.aovNormal[index]=make_float4(normal,0.0f);")
    o = _between(stmts(function_body(texts["orc_render"], "raygenPathTracer")), "if(!(std::isnan(radiance.x)", "o.aovNormal[index]=make_float4(normal,0.0f);")
    r = mapped(r, {
        "const unsigned int index=theLaunchIndex.y*theLaunchDim.x+theLaunchIndex.x;": None,   # computed earlier (test_raygen_path_tracer)
        "float4*bufferRGBA=reinterpret_cast<float4*>(sysData.outputBuffer);": None,
        "float4*bufferAlbedo=reinterpret_cast<float4*>(sysData.albedoBuffer);": None,
        "float4*bufferNormal=reinterpret_cast<float4*>(sysData.normalBuffer);": None,
        "radiance=lerp(make_float3(o.output[index]),radiance,t);": ["const float4 dst=o.output[index];", "radiance=lerp(make_float3(dst),radiance,t);", "if(o.aov)"],
        "o.output[index]=make_float4(radiance,1.0f);": ["o.output[index]=make_float4(radiance,1.0f);", "if(o.aov)"],
    })
    assert o == r, _diff(r, o)


def test_spherical_cdf_and_gaussian_filter(texts):
    """Texture::calculateSphericalCDF + gaussianFilter (apps/rtigo3/src/Texture.cpp:1499-1645): the one host piece the
    reference-built helper cannot execute (Texture.cpp needs DevIL and the CUDA driver)."""
    tex = read(os.path.join(REF_SRC, "Texture.cpp"))
    orc_text = texts["orc_render"]
    ref = stmts(function_body(tex, "gaussianFilter"))
    orc = stmts(function_body(orc_text, "gaussianFilter"))
    assert orc == ref, _diff(ref, orc)
    ref = stmts(function_body(tex, "Texture::calculateSphericalCDF"), REF_RENAMES)
    ref = mapped(ref, {
        "float*funcU=new float[m_width*m_height];": ["const Texture&tex=o.sys.textures[2];", "const unsigned int m_width=tex.width,m_height=tex.height;",
                                                      "const float*rgba=reinterpret_cast<const float*>(tex.texels);", "std::vector<float>funcU(m_width*m_height),funcV(m_height+1);"],
        "float*funcV=new float[m_height+1];": None,
        "m_integral=sum*2.0f*M_PIf_*M_PIf_/float(m_width*m_height);": "o.sys.envIntegral=sum*2.0f*M_PIf_*M_PIf_/float(m_width*m_height);",
        "float*cdfU=new float[(m_width+1)*m_height];": ["o.sys.envCDF_U.assign((m_width+1)*m_height,0.0f);", "o.sys.envCDF_V.assign(m_height+1,0.0f);", "float*cdfU=o.sys.envCDF_U;"],
        "float*cdfV=new float[m_height+1];": "float*cdfV=o.sys.envCDF_V;",
        "funcV[m_height]=integral;": None,   # "For completeness, actually unused." (Texture.cpp:1613)
        # upload and clean-up: the tables stay in the oracle's SystemData
        "size_t sizeBytes=(m_width+1)*m_height*sizeof(float);": None, "CU_CHECK(cuMemAlloc(&m_d_envCDF_U,sizeBytes));": None, "CU_CHECK(cuMemcpyHtoD(m_d_envCDF_U,cdfU,sizeBytes));": None,
        "sizeBytes=(m_height+1)*sizeof(float);": None, "CU_CHECK(cuMemAlloc(&m_d_envCDF_V,sizeBytes));": None, "CU_CHECK(cuMemcpyHtoD(m_d_envCDF_V,cdfV,sizeBytes));": None,
        "delete[]cdfV;": None, "delete[]cdfU;": None, "delete[]funcV;": None, "delete[]funcU;": ["o.sys.envWidth=m_width;", "o.sys.envHeight=m_height;"],
    })
    orc = stmts(function_body(orc_text, "calculateSphericalCDF"), extra_spelling=CASTS + [(r"\.data\(\)", "")])
    assert orc == ref, _diff(ref, orc)


def test_screenshot_tonemap_loop(texts):
    """Application::screenshot, tonemap branch (apps/rtigo3/src/Application.cpp:2259-2297) vs screenshotTonemap."""
    app = read(os.path.join(REF_SRC, "Application.cpp"))
    ren = [(r"fmaxf\(make_float3\(0\.0f\),", "fmaxf3v(make_float3(0.0f),"), (r"\bpowf\(ldrColor,", "powf3(ldrColor,"), (r"\bclamp\(powf3\(", "clamp3(powf3(")]
    ref = stmts(function_body(app, "Application::screenshot"), ren)
    ref = _between(ref, "const float invGamma=", "dst[idx]=make_uchar3(")
    ref = mapped(ref, {   # the x / y loops over the image, flattened
        "for(int y=0; y<m_resolution.y;++y)": "for(size_t idx=0; idx<numPixels;++idx)",
        "for(int x=0; x<m_resolution.x;++x)": None,
        "const int idx=y*m_resolution.x+x;": None,
    })
    orc = stmts(function_body(texts["orc_render"], "screenshotTonemap"))
    assert orc == ref, _diff(ref, orc)


# =====================================================================================================================
# Round 5: the reference's compile-time lighting switch and its debug filter (shaders/config.h:50-56). The oracle carries
# the same #if blocks; the programs they touch are compared again with each switch flipped.
# =====================================================================================================================
def test_next_event_estimation_off():
    """USE_NEXT_EVENT_ESTIMATION 0: closesthit.cu:202-214,250-304, miss.cu:62-68,92-106, Optix7Gui closesthit.cu:202-214."""
    with flipped(USE_NEXT_EVENT_ESTIMATION=0):
        t = load_texts()
        ref = statements(function_body(t["closesthit.cu"], "__closesthit__radiance"), REF_RENAMES)
        assert not any("lightSample" in x or "powerHeuristic" in x for x in ref), "the reference's brute-force build samples no light"
        orc = statements(function_body(t["orc_render"], "closesthitRadiance"))
        assert not any("lightSample" in x or "powerHeuristic" in x or "traceShadow" in x for x in orc)
        test_closesthit_radiance(t)
        test_miss_programs(t)
        test_integrator_loop(t)
        test_optix7gui_light_block(t)


def test_debug_exceptions_on():
    """USE_DEBUG_EXCEPTIONS 1: raygeneration.cu:205-218 — the false-colour filter replaces the NaN filter."""
    with flipped(USE_DEBUG_EXCEPTIONS=1):
        t = load_texts()
        ref = stmts(function_body(t["raygeneration.cu"], "__raygen__path_tracer"), RAYGEN_RENAMES, CASTS)
        assert "radiance=make_float3(0.0f,0.0f,1000000.0f);" in ref
        test_raygen_path_tracer(t)
