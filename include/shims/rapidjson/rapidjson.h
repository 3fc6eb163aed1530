// rapidjson/rapidjson.h — see rapidjson/document.h (independent minimal stand-in).
#pragma once
