// rapidjson/error/en.h — English text for the parse errors of the stand-in DOM (rapidjson/document.h).
#pragma once

#include "../document.h"

namespace rapidjson {
inline const char * GetParseError_En(ParseErrorCode code)
{
    switch (code) {
    case kParseErrorNone: return "No error.";
    case kParseErrorDocumentEmpty: return "The document is empty.";
    case kParseErrorDocumentRootNotSingular: return "The document root must not be followed by other values.";
    case kParseErrorValueInvalid: return "Invalid value.";
    case kParseErrorObjectMissName: return "Missing a name for object member.";
    case kParseErrorObjectMissColon: return "Missing a colon after a name of object member.";
    case kParseErrorObjectMissCommaOrCurlyBracket: return "Missing a comma or '}' after an object member.";
    case kParseErrorArrayMissCommaOrSquareBracket: return "Missing a comma or ']' after an array element.";
    case kParseErrorStringMissQuotationMark: return "Missing a closing quotation mark in string.";
    case kParseErrorTermination: return "Terminate parsing due to unexpected end of input.";
    }
    return "Unknown error.";
}
}  // namespace rapidjson
