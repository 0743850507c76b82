// Shared between the translation units of libumpc_mi355x.so: the message umpcLastError() returns.
#ifndef UMPC_ERR_H
#define UMPC_ERR_H
__attribute__((visibility("hidden"))) void umpc_set_error(const char *msg);
#endif
