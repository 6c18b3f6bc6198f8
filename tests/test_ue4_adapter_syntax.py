"""integration/ue4/OctreeSearch.{h,cpp} — the UE4 adapter, this repository's own code — through a compiler's parser once.

Unreal Engine 4.9 and UnrealHeaderTool are not in the image, so the adapter cannot be BUILT here.  tests/cpp/ue4_syntax/ declares
the minimum of the engine's names the two files use (AActor, TArray, FVector, FColor, DrawDebug*, the UHT macros as no-ops) and
`g++ -fsyntax-only` reads the adapter against them and against the real include/nbody_actor.hpp.  It pins no arithmetic, is no
oracle and no build of the reference (tests/cpp/ue4_syntax/README.md); it catches typos, missing members and signature drift.
Round 5's first run of it found one real defect: include/nbody.h defined NBODY_API, the very macro UnrealBuildTool defines for the
reference's module NBody (`class NBODY_API AOctreeSearch`, OctreeSearch.h:112) — now NBODY_AMD_API."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECLS = os.path.join(ROOT, "tests", "cpp", "ue4_syntax")
UE4 = os.path.join(ROOT, "integration", "ue4")


def _syntax_only(source, *extra):
    return subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", DECLS,
                           "-I", os.path.join(ROOT, "include"), "-I", UE4, *extra, source], capture_output=True, text=True, timeout=120)


def test_the_adapter_parses_against_the_declarations_it_needs():
    out = _syntax_only(os.path.join(UE4, "OctreeSearch.cpp"))
    assert out.returncode == 0, out.stderr


def test_the_parser_really_reads_the_adapter(tmp_path):
    # (the check must be able to fail: the same file with one member misspelt does not parse)
    text = open(os.path.join(UE4, "OctreeSearch.cpp")).read()
    assert "Engine.PushParticles();" in text
    broken = tmp_path / "OctreeSearch.cpp"
    broken.write_text(text.replace("Engine.PushParticles();", "Engine.PushParticle();"))
    out = _syntax_only(str(broken))
    assert out.returncode != 0 and "PushParticle" in out.stderr


def test_the_library_headers_leave_the_host_modules_export_macro_alone():
    # UnrealBuildTool defines <MODULE>_API for every module; the reference's module is NBody (NBody.Build.cs, OctreeSearch.h:112)
    for name in os.listdir(os.path.join(ROOT, "include")):
        text = open(os.path.join(ROOT, "include", name)).read()
        assert not re.search(r"#\s*define\s+NBODY_API\b", text), name


def test_the_adapter_keeps_every_member_of_the_reference_class():
    # OctreeSearch.h:111-149: the members and UFUNCTIONs Blueprints bind to, by name
    text = open(os.path.join(UE4, "OctreeSearch.h")).read()
    for member in ("float Size;", "TArray<FParticle> Particles;", "Octree* ParticleOctree;", "bool Initialized;", "bool ShowOctree;",
                   "float PhDeltaTime;", "AOctreeSearch();", "virtual void BeginPlay() override;", "virtual void Tick(float DeltaSeconds) override;",
                   "void DrawOctreeBoxes(Octree* Oct);", "void ComputeCubeSize();", "void CreateSpacePoints(int32 N, float Size = 200);",
                   "void CreateOctree();", "void CleanParticles();"):
        assert member in text, member
    assert "class NBODY_API AOctreeSearch : public AActor" in text
