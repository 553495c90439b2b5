#include "opencv.hpp"
