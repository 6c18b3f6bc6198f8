// Minimum declarations for `g++ -fsyntax-only` of integration/ue4/OctreeSearch.{h,cpp} — see README.md next to this directory.
// Signatures only, as the adapter uses them (UE4 4.9 API from memory); nothing here is engine code, an oracle, or arithmetic.
#pragma once
#include <cstddef>
#include <cstdint>

typedef int32_t int32;
typedef uint8_t uint8;

// UnrealHeaderTool's markers: nothing for a plain compiler
#define USTRUCT(...)
#define UCLASS(...)
#define UPROPERTY(...)
#define UFUNCTION(...)
#define GENERATED_BODY(...)
#define GENERATED_USTRUCT_BODY(...)
#define NBODY_API             /* UnrealBuildTool's export macro of the module NBody (OctreeSearch.h:112): a header of the library must leave it alone */

struct FVector {
  float X, Y, Z;
  FVector();
  FVector(float InX, float InY, float InZ);
  static const FVector ZeroVector;
};

struct FColor {
  uint8 R, G, B, A;
  static const FColor Red, Black;
};

template <typename T> class TArray {
 public:
  int32 Num() const;
  T *GetData();
  const T *GetData() const;
  void SetNumUninitialized(int32 NewNum);
  void Empty();
  T &operator[](int32 Index);
};

class UWorld;

struct FActorTickFunction { bool bCanEverTick; };

class AActor {
 public:
  virtual ~AActor();
  virtual void BeginPlay();
  virtual void Tick(float DeltaSeconds);
  UWorld *GetWorld() const;
  FVector GetActorLocation() const;
  FActorTickFunction PrimaryActorTick;
};

// what UnrealHeaderTool's GENERATED_BODY gives a UCLASS: the base class under the name Super
#define NBODY_SYNTAX_SUPER(Base) typedef Base Super;
