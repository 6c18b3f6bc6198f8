// The reference module's precompiled header by name (Source/NBody/NBody.h includes Engine.h); nothing of it is needed to parse the adapter.
#pragma once
