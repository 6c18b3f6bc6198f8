// Signatures of the three debug-draw calls the adapter makes (OctreeSearch.cpp:24, 40, 41 in the reference), for the parser only.
#pragma once
#include "GameFramework/Actor.h"
void FlushPersistentDebugLines(const UWorld *InWorld);
void DrawDebugBox(const UWorld *InWorld, FVector const &Center, FVector const &Extent, FColor const &Color, bool bPersistentLines = false,
                  float LifeTime = -1.f, uint8 DepthPriority = 0);
void DrawDebugPoint(const UWorld *InWorld, FVector const &Position, float Size, FColor const &PointColor, bool bPersistentLines = false,
                    float LifeTime = -1.f, uint8 DepthPriority = 0);
